// rpl_voxel.hip — k_cloud_voxel: raw scan -> clipped, voxel-downsampled PointCloud2
// (extensions E1 + E2 + E4 of SURVEY.md §8 a-ext) in ONE streaming pass over the packed
// 8-byte nodes (reference layout src/sdk/include/sl_lidar_cmd.h:272-278).  One 1024-thread
// workgroup (16 wave64) owns one scan.
//
// The kernel is VALU-issue bound, not latency bound (profiles/r01): on gfx950 only the plain
// VOP2 add/sub/mul/fma/and/or/lshr/mov forms issue in ~2.4 cycles per wave, every other
// vector op (conversions, floor, compares, selects, DPP, packed fp32, v_mbcnt, VOP3) takes
// ~4.2, and ds_bpermute 24.  Phase S is therefore written to minimise issue cycles:
//
// Phase S (streaming, straight-line code):
//   * one global_load_dwordx4 per lane = TWO consecutive samples (A, B); the (cos, sin)
//     table entries of the next round are fetched one round ahead;
//   * keep mask = one unsigned interval test on dist_mm_q2 (host-derived, rpl_device.hpp);
//   * x/y arithmetic on packed fp32 pairs (v_pk_mul_f32 / v_pk_fma_f32): polar->XY, the two
//     validated mul+FMA divides by the leaf, the exact in-cell remainder;
//   * cell key without integer conversions: (floor + 2^23 + 32768) puts iy/ix + 32768 into
//     the low mantissa bits, one v_perm_b32 packs (iy, ix) into the 32-bit sort key;
//   * a smooth ring stays in a 5 cm cell for ~10-200 samples, so runs of equal keys are
//     aggregated before anything is stored: A and B merge in the lane, lanes merge through
//     three plain DPP prefix scans, and a lane whose run ends writes ONE 16-byte record
//     {key, prefix_x, prefix_y, prefix_count|intensity|tag} to an LDS queue.  The record
//     holds the wave-pass PREFIX, not the run sum: the run sum is prefix(this record) -
//     prefix(previous record of the same wave-pass), recovered in phase R, which removes
//     every cross-lane gather (ds_bpermute) and every segmented-scan mask from the hot loop.
// Phase R (per scan, regular data-parallel passes over the <= 7168 run records):
//   prefix -> run sums, counting sort by row + rank inside the row -> records in (iy, ix)
//   order -> segmented integer sums over equal keys -> one output point per cell.
//
// Fixed point: offset = (x - ix*leaf) * 2^K + 2^15 with 2^-K = ulp(leaf) (K = 28 for
// 5 cm).  One fp32 FMA yields x - ix*leaf EXACTLY whenever x is a multiple of 2^-K
// (|x| >= 3 cm at K = 28), so the integer sums are exact and sum/count reproduces the
// spec's fp64 running sum bit for bit; closer to the axes the per-point error is
// <= 2^-K m (3.7e-9 m), far below the 1e-6 m bar.  Integer sums are order
// independent, so the kernel is run-to-run deterministic although the queue order is not.
//
// A scan whose records do not fit (or that spans > 2048 rows) is processed in key
// bands: the key range is bisected until a band fits, each band re-streaming the scan
// (from L2 / Infinity Cache).
#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

#ifndef RPL_RECCAP
#define RPL_RECCAP 7168
#endif
#ifndef RPL_ROWCAP
#define RPL_ROWCAP 2048
#endif
constexpr uint32_t kRecCap = RPL_RECCAP;              // run records per scan (112 KiB)
constexpr uint32_t kRecPerThread = kRecCap / kBlock;  // 7
constexpr uint32_t kRowCap = RPL_ROWCAP;              // rows the counting sort handles
constexpr uint32_t kEmptyKey = 0xFFFFFFFFu;
constexpr float kKeyMagic = 8421376.0f;               // 2^23 + 32768

typedef float f2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int CTRL, int ROWMASK>
__device__ __forceinline__ uint32_t dpp_add(uint32_t v) {
  return v + (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, false);
}
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
// single inclusive wave64 prefix sum (phase R, not on the hot path)
__device__ __forceinline__ uint32_t wave_incl_scan_dpp(uint32_t v) {
  v = dpp_add<0x111, 0xF>(v);  // row_shr:1
  v = dpp_add<0x112, 0xF>(v);  // row_shr:2
  v = dpp_add<0x114, 0xF>(v);  // row_shr:4
  v = dpp_add<0x118, 0xF>(v);  // row_shr:8
  v = dpp_add<0x142, 0xA>(v);  // row_bcast:15 -> rows 1,3
  v = dpp_add<0x143, 0xC>(v);  // row_bcast:31 -> rows 2,3
  return v;
}

struct VoxelLds {
  uint4 rec[kRecCap];  // S: {key, prefix_x, prefix_y, tag<<24 | prefix(count<<16 | intensity)}
                       // R: {key, sum_x, sum_y, count<<16 | intensity_sum}
  uint32_t rowstart[kRowCap];
  uint32_t rowfill[kRowCap];
  alignas(16) uint32_t bucket[kRecCap + 8];  // (ix << 16 | record index) grouped by row
  uint32_t band_lo[34], band_hi[34];
  uint32_t misc[16];  // 0 queue tail, 1 status, 2 overflow, 3 sp, 4 rowmin, 5 rowmax, 7 out_base
  uint32_t tmp[32];
  double rcp[256];  // RN(1/count) for count < 256 (copied once per workgroup from the host table)
};

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

// Per-sample arithmetic of phase S for one 8-byte node (lo, hi) and its table entry `c`.
// Outputs the sort key (kEmptyKey when the sample is dropped) and the three quantities
// that are summed per cell.  `flags` collects RPLGPU_SCAN_CELL_RANGE.
template <bool FAST_DIV, bool BAND, bool SAFE, bool HASQ>
__device__ __forceinline__ bool voxel_sample(uint32_t lo, uint32_t hi, float2 c, const KParams &p,
                                             uint32_t q_min16, uint32_t ibfe_off,
                                             uint32_t ibfe_w, uint32_t klo, uint32_t khi,
                                             uint32_t &key, uint32_t &qx, uint32_t &qy,
                                             uint32_t &ci, uint32_t &flags) {
  // Straight-line: a dropped sample runs the same arithmetic on harmless operands (dist 0 or an
  // out-of-range distance give finite values) and is masked at the end.  A wave issues in
  // order, so the exec-mask regions and branches of an `if (kept)` cost it more than the few
  // instructions they skip on the ~10 % of dropped samples (profiles/r01).
  const uint32_t d = __builtin_amdgcn_alignbit(hi, lo, 16);  // unaligned u32 at byte 2
  bool kept = (d - p.d_lo) <= p.d_span;                      // E1 (and :584)
  if (HASQ) kept = kept & ((hi & 0x00FF0000u) >= q_min16);
  const float df = __uint2float_rn(d);
  const float dm = FAST_DIV ? div_by(df, 4000.0f, 0.00025f) : df / 4000.0f;  // :590
  const f2 cv = {c.x, c.y};
  const f2 xy = cv * dm;                                                       // E2
  f2 t;
  if (FAST_DIV) {
    t = div_by2(xy, p.voxel_leaf, p.inv_leaf);                                 // E4 cell
  } else {
    t.x = xy.x / p.voxel_leaf;
    t.y = xy.y / p.voxel_leaf;
  }
  const f2 f = {__builtin_floorf(t.x), __builtin_floorf(t.y)};
  if (!SAFE) {
    const bool inr = (fabsf(f.x) < 32767.0f) && (fabsf(f.y) < 32767.0f);
    if (kept && !inr) flags |= RPLGPU_SCAN_CELL_RANGE;
    kept = kept & inr;
  }
  // iy + 32768 | ix + 32768 from the mantissas of f + (2^23 + 32768)
  const uint32_t kx = __float_as_uint(f.x + kKeyMagic);
  const uint32_t ky = __float_as_uint(f.y + kKeyMagic);
  const uint32_t k = __builtin_amdgcn_perm(ky, kx, 0x05040100u);
  if (BAND) kept = kept & ((k - klo) <= (khi - klo));
  const f2 lf = {p.voxel_leaf, p.voxel_leaf};
  const f2 r = __builtin_elementwise_fma(-f, lf, xy);  // x - ix*leaf, exact
  const f2 o = r * p.vox_scale_f;
  const uint32_t m = kept ? 0xFFFFFFFFu : 0u;
  key = kept ? k : kEmptyKey;
  qx = (uint32_t)((int)o.x + p.vox_bias) & m;
  qy = (uint32_t)((int)o.y + p.vox_bias) & m;
  ci = ((1u << 16) | ((hi >> ibfe_off) & ibfe_w)) & m;  // count | intensity (:591-592)
  return kept;
}

// Cross-lane part of one wave-pass over 128 samples: lane l holds samples A = 2l, B = 2l+1
// (okA / okB: the sample survived the keep mask and carries a real key).
// Returns false when the record queue is full (wave-uniform).
__device__ __forceinline__ bool voxel_pair_pass(VoxelLds &L, uint32_t tag, bool okA, uint32_t keyA,
                                                uint32_t xA, uint32_t yA, uint32_t cA, bool okB,
                                                uint32_t keyB, uint32_t xB, uint32_t yB,
                                                uint32_t cB) {
#ifdef RPL_ABL_NOAPPEND
  asm volatile("" ::"v"(keyA), "v"(xA), "v"(yA), "v"(cA), "v"(keyB), "v"(xB), "v"(yB), "v"(cB));
  return true;
#endif
  // lane l+1's first key; lane 63 sees a value no key can take (its run always ends)
  const uint32_t nextA = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFEu, (int)keyA, 0x130,
                                                               0xF, 0xF, false);  // wave_shl:1
  // run-end masks built in scalar registers: (sample kept) & (key differs from the next one).
  // Written with explicit compares because `ballot(a && b)` is materialised by the compiler as
  // v_cndmask + v_cmp per mask; the per-lane predicates for the stores come back from the
  // masks with inverse_ballot (one s_and_saveexec each).
  uint64_t ne1, ne2;
  asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(ne1) : "v"(keyA), "v"(keyB));
  asm("v_cmp_ne_u32_e64 %0, %1, %2" : "=s"(ne2) : "v"(keyB), "v"(nextA));
  const uint64_t m1 = __builtin_amdgcn_ballot_w64(okA) & ne1;  // a run ends at A
  const uint64_t m2 = __builtin_amdgcn_ballot_w64(okB) & ne2;  // a run ends at B
  const bool e1 = __builtin_amdgcn_inverse_ballot_w64(m1);
  const bool e2 = __builtin_amdgcn_inverse_ballot_w64(m2);
  const uint32_t total = (uint32_t)__popcll(m1) + (uint32_t)__popcll(m2);
  if (total == 0u) return true;  // wave-uniform: nothing kept in this pass
  // reserve queue slots: one LDS atomic by lane 0, its round trip overlaps the scans below
  // (hand-placed so that the compiler's atomic optimiser does not wait for it right away)
  uint32_t base = 0u;
#ifndef RPL_ABL_NOATOMIC
  if (lane_id() == 0) {
    asm volatile("ds_add_rtn_u32 %0, %1, %2"
                 : "=v"(base)
                 : "v"((uint32_t)(uintptr_t)&L.misc[0]), "v"(total)
                 : "memory");
  }
#endif
  uint32_t Px = xA + xB, Py = yA + yB, Pc = cA + cB;
#ifndef RPL_ABL_NOSCAN
  wave_incl_scan3_dpp(Px, Py, Pc);
#endif
#ifndef RPL_ABL_NOATOMIC
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(base)::"memory");
#endif
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (base + total > kRecCap) return false;  // wave-uniform: queue full -> bisect the band
  const uint32_t mb1 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m1 >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)m1, 0u));
  const uint32_t mb2 = __builtin_amdgcn_mbcnt_hi((uint32_t)(m2 >> 32),
                                                 __builtin_amdgcn_mbcnt_lo((uint32_t)m2, 0u));
  const uint32_t pos1 = base + mb1 + mb2;  // records are queued in sample order
  uint32_t pos2;                           // pos1 + (e1 ? 1 : 0): the ballot is the carry-in
  uint64_t carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(pos2), "=s"(carry_out) : "v"(pos1), "s"(m1));
  if (e1) L.rec[pos1] = make_uint4(keyA, Px - xB, Py - yB, (Pc - cB) | tag);
  if (e2) L.rec[pos2] = make_uint4(keyB, Px, Py, Pc | tag);
  return true;
}

// ------------------------------------------------------------------------------
// Phase R for one key band: the queue of run records -> output cells in (iy, ix) order.
// Returns 0 = done (ncell written to *ncell_out), 1 = the band must be bisected (too many rows).
// ------------------------------------------------------------------------------
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
};
enum : int { kEmitLegacy = 0, kEmitArenaFirst = 1, kEmitCountOnly = 2, kEmitArenaKnown = 3 };

__device__ __forceinline__ uint32_t voxel_reduce(VoxelLds &L, const KParams &p, const double *rcp,
                                              float4 *__restrict__ out, uint32_t out_stride,
                                              uint32_t b, uint32_t *ncell_out, int mode,
                                              const VoxelArena &arena) {
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = clock64();
#define RPL_MARK(i)                      \
  {                                      \
    unsigned long long now_ = clock64(); \
    tacc[i] += now_ - t_mark;            \
    t_mark = now_;                       \
  }
  auto flush_dbg = [&]() {
    if (p.dbg && threadIdx.x == 0) {
#pragma unroll
#if defined(RPL_ABL_WAITT) || defined(RPL_ABL_LAT)
      for (int i = 3; i < 8; ++i) atomicAdd(&p.dbg[8 * b + i], tacc[i]);
#else
      for (int i = 1; i < 8; ++i) atomicAdd(&p.dbg[8 * b + i], tacc[i]);
#endif
    }
  };
  const int vbias = p.vox_bias;
  const uint32_t nrec = L.misc[0];
  if (p.dbg && threadIdx.x == 0) tacc[7] += (unsigned long long)nrec << 40;  // developer aid
  // thread t owns the queue records t, t + 1024, ... (a wave's seven 64-record slices come
  // from seven different parts of the scan, which balances the ranking work below);
  // prefix -> run sum against the record just before it when both come from the same
  // wave-pass (equal tags)
  uint4 mine[kRecPerThread];
  uint32_t rmin = 0xFFFFFFFFu, rmax = 0u;
#pragma unroll
  for (int k = 0; k < (int)kRecPerThread; ++k) {
    const uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
    uint4 m = make_uint4(kEmptyKey, 0u, 0u, 0u);
    // whole 64-record slices beyond the queue tail are skipped (wave-uniform: the mean queue
    // holds 3.1 k of 7168 records); the previous record comes from the lane to the left
    if ((idx & ~63u) < nrec) {
      const bool ok = idx < nrec;
      const uint4 raw = L.rec[idx];  // idx < kRecCap always
      uint4 pr;
      pr.y = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.y, 0x138, 0xF, 0xF, false);
      pr.z = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.z, 0x138, 0xF, 0xF, false);
      pr.w = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)raw.w, 0x138, 0xF, 0xF, false);
      if (lane_id() == 0u && idx > 0u) {
        const uint4 t = L.rec[idx - 1u];
        pr.y = t.y; pr.z = t.z; pr.w = t.w;
      }
      const bool same = ok && idx > 0u && (((raw.w ^ pr.w) >> 24) == 0u);
      m = raw;
      m.y -= same ? pr.y : 0u;
      m.z -= same ? pr.z : 0u;
      m.w = (raw.w & 0x00FFFFFFu) - (same ? (pr.w & 0x00FFFFFFu) : 0u);
      if (!ok) m = make_uint4(kEmptyKey, 0u, 0u, 0u);
      if (ok) {
        rmin = min(rmin, m.x >> 16);
        rmax = max(rmax, m.x >> 16);
        atomicAdd(&L.rowstart[(m.x >> 16) & (kRowCap - 1u)], 1u);  // counting sort over rows
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
  __syncthreads();  // every prefix was read before any record slot is rewritten below
  RPL_MARK(1)
  rmin = L.misc[4];
  const uint32_t out_base = L.misc[7];
  uint32_t ncell = 0;
  if (rmin != 0xFFFFFFFFu) {  // at least one record (block-uniform)
    if (L.misc[5] - rmin + 1u > kRowCap) {
      flush_dbg();
      return 1u;
    }
    RPL_MARK(2)
    {  // exclusive scan over the kRowCap rows in row order (2 per thread); a row iy lives at
       // iy mod kRowCap, so scan position j is the physical row (j + rmin) mod kRowCap
      const uint32_t p0 = (2u * threadIdx.x + rmin) & (kRowCap - 1u);
      const uint32_t p1 = (2u * threadIdx.x + 1u + rmin) & (kRowCap - 1u);
      uint32_t r0 = L.rowstart[p0], r1 = L.rowstart[p1];
      uint32_t tot;
      uint32_t ex = block_excl_scan(r0 + r1, L.tmp, &tot);
      L.rowstart[p0] = ex;
      L.rowstart[p1] = ex + r0;
      L.rowfill[p0] = ex;
      L.rowfill[p1] = ex + r0;
    }
    __syncthreads();
    RPL_MARK(3)
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      if (mine[k].x != kEmptyKey) {
        uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
        uint32_t pos = atomicAdd(&L.rowfill[(mine[k].x >> 16) & (kRowCap - 1u)], 1u);
        L.bucket[pos] = (mine[k].x << 16) | idx;  // (ix, record index): unique
      }
    }
    __syncthreads();
    RPL_MARK(4)
    // rank inside the row = number of smaller entries of the same row, then permute the
    // records in place (they are all in registers; nobody reads rec now)
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      if (mine[k].x != kEmptyKey) {
        const uint32_t idx = threadIdx.x + (uint32_t)k * kBlock;
        const uint32_t row = (mine[k].x >> 16) & (kRowCap - 1u);
        const uint32_t me = (mine[k].x << 16) | idx;
        const uint32_t s0 = L.rowstart[row], s1 = L.rowfill[row];
        // the row segment [s0, s1) as aligned 16-byte blocks: the first and the last block are
        // compared under position masks, the ones between them as they are; the first six
        // reads are issued together (one LDS round trip covers rows of up to 24 records)
        const uint32_t a0 = s0 & ~3u, e0 = (s1 + 3u) & ~3u;  // s1 > s0: this record is in it
        uint32_t rank = s0;
        {
          const uint4 v = *reinterpret_cast<const uint4 *>(&L.bucket[a0]);
          rank += (a0 >= s0 && a0 < s1 && v.x < me) + (a0 + 1u >= s0 && a0 + 1u < s1 && v.y < me) +
                  (a0 + 2u >= s0 && a0 + 2u < s1 && v.z < me) + (a0 + 3u < s1 && v.w < me);
        }
        if (e0 - a0 > 4u) {
          const uint32_t l0 = e0 - 4u;  // last block: l0 >= a0 + 4 > s0
          const uint4 w = *reinterpret_cast<const uint4 *>(&L.bucket[l0]);
          rank += (w.x < me) + (l0 + 1u < s1 && w.y < me) + (l0 + 2u < s1 && w.z < me) +
                  (l0 + 3u < s1 && w.w < me);
#pragma unroll
          for (uint32_t j = 1; j <= 4u; ++j) {
            if (a0 + 4u * j < l0) {
              const uint4 v = *reinterpret_cast<const uint4 *>(&L.bucket[a0 + 4u * j]);
              rank += (v.x < me) + (v.y < me) + (v.z < me) + (v.w < me);
            }
          }
          for (uint32_t m = a0 + 20u; m < l0; m += 16u) {  // long rows: four blocks per trip
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
              if (m + 4u * j < l0) {
                const uint4 v = *reinterpret_cast<const uint4 *>(&L.bucket[m + 4u * j]);
                rank += (v.x < me) + (v.y < me) + (v.z < me) + (v.w < me);
              }
            }
          }
        }
        L.rec[rank] = mine[k];
      }
    }
    __syncthreads();
    RPL_MARK(5)
    // heads of equal-key groups -> cell index; thread t owns sorted records [7t, 7t+7)
    const uint32_t r_lo = threadIdx.x * kRecPerThread;
    uint32_t headbits = 0, nheads = 0;
    uint32_t prevkey = (r_lo > 0 && r_lo <= nrec) ? L.rec[r_lo - 1].x : kEmptyKey;
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k) {
      uint32_t r = r_lo + k;
      uint32_t key = (r < nrec) ? L.rec[r].x : kEmptyKey;
      if (r < nrec && key != prevkey) {
        headbits |= 1u << k;
        ++nheads;
      }
      prevkey = key;
    }
    uint32_t cell = block_excl_scan(nheads, L.tmp, &ncell);
    RPL_MARK(6)
    // position of every cell's first record (the bucket array is free again)
#pragma unroll
    for (int k = 0; k < (int)kRecPerThread; ++k)
      if ((headbits >> k) & 1u) L.bucket[cell++] = r_lo + k;
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
      nemit = at >= arena.capacity ? 0u : (uint32_t)min((unsigned long long)ncell, arena.capacity - at);
    } else if (mode == kEmitCountOnly) {
      nemit = 0u;  // several bands: first learn the total, the cells are written in a second go
    }
    for (uint32_t c = threadIdx.x; c < nemit; c += kBlock) {
      uint32_t r = L.bucket[c];
      // a cell is 1.13 records on average: fetch the head and the two records behind it in one
      // round trip, continue serially only when all three belong to the cell
      const uint4 q0 = L.rec[r];
      const uint4 q1 = L.rec[min(r + 1u, kRecCap - 1u)];
      const uint4 q2 = L.rec[min(r + 2u, kRecCap - 1u)];
      const uint32_t key = q0.x;
      const bool m1 = (r + 1u < nrec) && (q1.x == key);
      const bool m2 = m1 && (r + 2u < nrec) && (q2.x == key);
      double sx = (double)q0.y, sy = (double)q0.z;
      // count and intensity sum are packed per RECORD (<= 128 samples: 16 bits each suffice);
      // per CELL they are summed apart -- a large or close cell collects thousands of samples
      // and its intensity sum passes 2^16 (found by tests/test_gpu_fuzz.py)
      uint32_t cnt = q0.w >> 16, isum = q0.w & 0xFFFFu;
      if (m1) { sx += (double)q1.y; sy += (double)q1.z; cnt += q1.w >> 16; isum += q1.w & 0xFFFFu; }
      if (m2) {
        sx += (double)q2.y; sy += (double)q2.z; cnt += q2.w >> 16; isum += q2.w & 0xFFFFu;
        for (r += 3u; r < nrec; ++r) {  // segmented sum over the rest of this cell's records
          const uint4 q = L.rec[r];
          if (q.x != key) break;
          sx += (double)q.y;
          sy += (double)q.z;
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
      out[out_base + c] = make_float4((float)(qx * inv_scale), (float)(qy * inv_scale), 0.0f,
                                      (float)qi);
    }
  }
  __syncthreads();
  RPL_MARK(7)
  flush_dbg();
#undef RPL_MARK
  *ncell_out = ncell;
  return 0u;
}

template <bool FAST_DIV, bool SAFE>
__global__ __launch_bounds__(kBlock) void k_cloud_voxel(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, Tables T, const uint32_t *__restrict__ keepmask, uint32_t mask_stride,
    float4 *__restrict__ xyzi, uint32_t out_stride, uint32_t *__restrict__ n_points,
    uint32_t *__restrict__ status, uint32_t B, VoxelArena arena) {
  __shared__ VoxelLds L;

  if (threadIdx.x < 256) L.rcp[threadIdx.x] = T.rcp[threadIdx.x];  // (first barrier below publishes it)
#ifdef RPL_ABL_CLK
  const unsigned long long clk_c0 = clock64(), clk_w0 = wall_clock64();
#endif
  // persistent workgroups: one per CU (a workgroup needs the whole LDS of a CU, so launching
  // one per scan only adds 4096 dispatches); the first scan is blockIdx.x, the next ones come
  // from a shared counter, so a workgroup that drew cheap scans simply takes more of them
  for (uint32_t b = blockIdx.x; b < B;) {
  const uint32_t n = min(n_per_scan[b], min(n_stride, kMaxN));  // never past the slot
  const uint2 *scan = nodes + (size_t)b * n_stride;
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

  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  const uint32_t ishift = p.is_new_protocol ? 0u : 2u;  // :591-592
  const uint32_t ibfe_off = 16u + ishift, ibfe_w = 0xFFu >> ishift;  // shift, mask
  const uint32_t q_min16 = p.clip_enable ? (min(p.q_min, 256u) << 16) : 0u;
  uint32_t flags = 0;
  unsigned long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t_mark = clock64();
#define RPL_MARK(i)                      \
  {                                      \
    unsigned long long now_ = clock64(); \
    tacc[i] += now_ - t_mark;            \
    t_mark = now_;                       \
  }

  // lane l of a round owns the sample pair (2i, 2i+1), i = round*1024 + thread
  const uint32_t npairs = (n + 1u) >> 1;
  // Bounds-checked buffer resource over this scan's n*8 bytes: a pair (or its second node)
  // beyond the scan reads as zero, i.e. dist 0, which the keep test drops.  Every lane
  // always issues the load, so the compiler's vmcnt bookkeeping is exact and the prefetch
  // distance below is really kept.
  const __amdgpu_buffer_rsrc_t scan_rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)scan, 0, (int)(n * 8u), 0x00020000);
  auto load_pair = [&](uint32_t i) -> uint4 {
#ifdef RPL_ABL_NORAW
    return make_uint4((i * 4u) & 0xFFFFu | ((i * 40000u) << 16), (i * 40000u) >> 16 | 0x00400000u,
                      (i * 4u + 2u) & 0xFFFFu | ((i * 40000u + 7u) << 16), (i * 40000u) >> 16 | 0x00800000u);
#else
#ifdef RPL_ABL_GLOBAL
    const uint32_t ii = min(i, (n >> 1) - 1u);
    return reinterpret_cast<const uint4 *>(scan)[ii];
#else
#ifndef RPL_RAW_AUX
#define RPL_RAW_AUX 2
#endif
    const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(scan_rsrc, (int)(i * 16u), 0, RPL_RAW_AUX);
    return make_uint4(t.x, t.y, t.z, t.w);
#endif
#endif
  };
#ifdef RPL_ABL_NOGATHER
  struct FakeCs { __device__ float2 operator[](uint32_t q) const { return make_float2(0.6f + (float)q * 1e-6f, 0.8f); } };
  const FakeCs cs_fake;
#define cs cs_fake
#endif

  const uint32_t *ror_bits = keepmask ? keepmask + (size_t)b * mask_stride : nullptr;
  bool first_band = true;
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
    }
    // the row histogram of this band is filled while the records are loaded in phase R (rows
    // are addressed modulo kRowCap, so the first row need not be known yet)
    for (uint32_t t = threadIdx.x; t < kRowCap; t += kBlock) L.rowstart[t] = 0u;
    __syncthreads();

    // ---- phase S: raw pairs two rounds ahead, table entries one round ahead -------------
    // (one loop instance per uniform condition, so that none of them is tested per pass)
    auto stream = [&](auto band_tag, auto hasq_tag, auto mask_tag) {
      constexpr bool BAND = decltype(band_tag)::value;
      constexpr bool HASQ = decltype(hasq_tag)::value;
      constexpr bool HASMASK = decltype(mask_tag)::value;
#ifdef RPL_ASM_RING
      // ---- streaming loop with hand-managed vector-memory counters -------------------------
      // Four raw-pair buffers W[r & 3] and two table buffers C[r & 1], addressed by name in a
      // four-round body, so nothing is ever copied.  Round r issues, in this order, the table
      // gathers of round r + 1 (their angle words came with raw buffer r + 1), [the E5 mask word
      // of round r + 1] and the raw pairs of round r + 3.  A wave's loads retire in order, so
      //   vmcnt(NB)     at the top    : raw r + 1 has arrived  (gathers r, raw r + 2 still fly)
      //   vmcnt(NB + 1) before the use: gathers r have arrived (raw r + 2, gathers r + 1, raw r + 3 fly)
      // with NB = 3 loads per round (4 with the mask).  A raw load has two rounds, a gather one
      // round to arrive.  The loads and waits are inline assembly because the compiler's own
      // counter tracking drains vmcnt to 0 at the loop header of any such ring (seen in the ISA
      // of the plain-HIP version), which costs more than the prefetch distance buys.  The waits
      // carry the loaded registers as operands, so no use can be scheduled above them, and
      // nothing but these statements touches the ring registers while a load is in flight
      // (checked in the ISA: no copy, no spill of W / C inside the loop).
      const uint32_t nrounds = (npairs + kBlock - 1u) / kBlock;
      const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)scan),
                        __builtin_amdgcn_readfirstlane((int)((uint32_t)((uintptr_t)scan >> 32) & 0xFFFFu)),
                        __builtin_amdgcn_readfirstlane((int)(n * 8u)), 0x00020000};
      const uint64_t cs_base = (uint64_t)(uintptr_t)cs;
      uint64_t mk_base = 0;
      if (HASMASK) {
        const uintptr_t mp = (uintptr_t)ror_bits;
        mk_base = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(mp >> 32)) << 32) |
                  (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)mp);
      }
      auto ld_raw = [&](u32x4 &w, uint32_t round_idx) {
        const uint32_t off = (round_idx * kBlock + threadIdx.x) * 16u;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen nt" : "=v"(w) : "v"(off), "s"(rs) : "memory");
      };
      auto ld_cs = [&](f2 &c, uint32_t word) {
        const uint32_t off = (word & 0xFFFFu) << 3;
        asm volatile("global_load_dwordx2 %0, %1, %2" : "=v"(c) : "v"(off), "s"(cs_base) : "memory");
      };
      auto ld_mask = [&](uint32_t &m, uint32_t round_idx) {
        const uint32_t pi = round_idx * kBlock + threadIdx.x;  // pair index -> bits 2*pi, 2*pi + 1
        const uint32_t off = min(pi >> 4, mask_stride - 1u) << 2;  // (past the scan: dist 0 anyway)
        asm volatile("global_load_dword %0, %1, %2" : "=v"(m) : "v"(off), "s"(mk_base) : "memory");
      };
      constexpr int NB = HASMASK ? 4 : 3;
      bool fits = true;  // wave-uniform: this wave has not seen the queue overflow
      auto round = [&](u32x4 &Wc, u32x4 &Wn, u32x4 &Wl, f2 &cA, f2 &cB, f2 &cAn, f2 &cBn,
                       uint32_t &mc, uint32_t &mn, uint32_t r) {
        if (NB == 3) asm volatile("s_waitcnt vmcnt(3)" : "+v"(Wn)::"memory");
        else asm volatile("s_waitcnt vmcnt(4)" : "+v"(Wn)::"memory");
        ld_cs(cAn, Wn.x);
        ld_cs(cBn, Wn.z);
        if (HASMASK) ld_mask(mn, r + 1u);
        ld_raw(Wl, r + 3u);
        if (NB == 3) asm volatile("s_waitcnt vmcnt(4)" : "+v"(cA), "+v"(cB), "+v"(Wc)::"memory");
        else asm volatile("s_waitcnt vmcnt(5)" : "+v"(cA), "+v"(cB), "+v"(mc), "+v"(Wc)::"memory");
        uint4 w = make_uint4(Wc.x, Wc.y, Wc.z, Wc.w);
        if (HASMASK) {  // a sample the E5 mask drops gets dist 0
          const uint32_t pi = r * kBlock + threadIdx.x;
          const uint32_t two = ((pi >> 4) < mask_stride) ? (mc >> ((pi & 15u) * 2u)) & 3u : 0u;
          if (!(two & 1u)) { w.x &= 0x0000FFFFu; w.y &= 0xFFFF0000u; }
          if (!(two & 2u)) { w.z &= 0x0000FFFFu; w.w &= 0xFFFF0000u; }
        }
        uint32_t keyA, xA, yA, ciA, keyB, xB, yB, ciB;
        const bool okA = voxel_sample<FAST_DIV, BAND, SAFE, HASQ>(
            w.x, w.y, make_float2(cA.x, cA.y), p, q_min16, ibfe_off, ibfe_w, klo, khi, keyA, xA, yA, ciA, flags);
        const bool okB = voxel_sample<FAST_DIV, BAND, SAFE, HASQ>(
            w.z, w.w, make_float2(cB.x, cB.y), p, q_min16, ibfe_off, ibfe_w, klo, khi, keyB, xB, yB, ciB, flags);
        // tag: (round, wave) — two neighbouring queue reservations never share it
        const uint32_t tag = (((r & 15u) << 4) | wave_id()) << 24;
        if (fits) fits = voxel_pair_pass(L, tag, okA, keyA, xA, yA, ciA, okB, keyB, xB, yB, ciB);
      };
      u32x4 W0, W1, W2, W3;
      f2 cA0, cB0, cA1, cB1;
      uint32_t m0 = 0xFFFFFFFFu, m1 = 0xFFFFFFFFu;
      ld_raw(W0, 0u);
      ld_raw(W1, 1u);
      asm volatile("s_waitcnt vmcnt(1)" : "+v"(W0)::"memory");
      ld_cs(cA0, W0.x);
      ld_cs(cB0, W0.z);
      if (HASMASK) ld_mask(m0, 0u);
      ld_raw(W2, 2u);
      for (uint32_t r = 0; r < nrounds && fits; r += 4u) {
        round(W0, W1, W3, cA0, cB0, cA1, cB1, m0, m1, r);
        if (r + 1u < nrounds) round(W1, W2, W0, cA1, cB1, cA0, cB0, m1, m0, r + 1u);
        if (r + 2u < nrounds) round(W2, W3, W1, cA0, cB0, cA1, cB1, m0, m1, r + 2u);
        if (r + 3u < nrounds) round(W3, W0, W2, cA1, cB1, cA0, cB0, m1, m0, r + 3u);
      }
      // loads requested past the last round are still in flight: their registers must not be
      // reused before they have landed
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(W0), "+v"(W1), "+v"(W2), "+v"(W3), "+v"(cA0), "+v"(cB0),
                   "+v"(cA1), "+v"(cB1), "+v"(m0), "+v"(m1)::"memory");
#else
#ifndef RPL_PASSES
#define RPL_PASSES 1
#endif
      // One loop trip = RPL_PASSES wave-passes (rounds).  All loads of the NEXT trip are issued at
      // the top of a trip — the table entries of trip t + 1 (their angle words arrived with the
      // raw pairs requested a trip earlier) and the raw pairs of trip t + 2 — so a load has a
      // whole trip of compute to arrive.  With one pass per trip that is ~900 cycles against a
      // measured ~1000 (raw) ... 1500 (raw behind the two gathers: a wave's loads return in
      // order) cycles of latency, and since the 16 waves of the workgroup run in lockstep nobody
      // computes while they all wait: 38 % of phase S was spent in that wait
      // (profiles/r02/voxel_wait_cycles.txt).  Two passes per trip hide it.
      constexpr int NP = RPL_PASSES;
#ifdef RPL_STAGGER_MASK
      // the 16 waves leave the barrier together and would issue their loads in one burst and
      // wait for them together; start them a fraction of a round apart instead
      for (uint32_t z = 0; z < (wave_id() & RPL_STAGGER_MASK); ++z) __builtin_amdgcn_s_sleep(RPL_STAGGER_SLEEP);
#endif
      uint4 w1[NP], w2[NP];
      float2 cA1[NP], cB1[NP];
#pragma unroll
      for (int j = 0; j < NP; ++j) w1[j] = load_pair((uint32_t)j * kBlock + threadIdx.x);
#pragma unroll
      for (int j = 0; j < NP; ++j) w2[j] = load_pair((uint32_t)(NP + j) * kBlock + threadIdx.x);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        cA1[j] = cs[w1[j].x & 0xFFFFu];
        cB1[j] = cs[w1[j].z & 0xFFFFu];
      }
      uint32_t round = 0;
#if defined(RPL_ABL_WAITT) || defined(RPL_ABL_LAT)
      unsigned long long wt_latch = 0;
#endif
      bool fits = true;  // wave-uniform: this wave has not seen the queue overflow
      for (uint32_t base = 0; base < npairs && fits; base += NP * kBlock) {
        uint4 w0[NP];
        float2 cA[NP], cB[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          w0[j] = w1[j];
          cA[j] = cA1[j];
          cB[j] = cB1[j];
          w1[j] = w2[j];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          cA1[j] = cs[w1[j].x & 0xFFFFu];
          cB1[j] = cs[w1[j].z & 0xFFFFu];
        }
#pragma unroll
        for (int j = 0; j < NP; ++j)
          w2[j] = load_pair(base + (uint32_t)(2 * NP + j) * kBlock + threadIdx.x);
#pragma unroll
        for (int j = 0; j < NP; ++j, ++round) {
          uint4 w = w0[j];
          if (HASMASK) {  // E5 mask (one bit per sample): a dropped sample gets dist 0
            const uint32_t pi = base + (uint32_t)j * kBlock + threadIdx.x;  // pair -> bits 2*pi, 2*pi+1
            const uint32_t word = pi >> 4;
            const uint32_t bits = (word < mask_stride) ? ror_bits[word] : 0u;
            const uint32_t two = (bits >> ((pi & 15u) * 2u)) & 3u;
            if (!(two & 1u)) { w.x &= 0x0000FFFFu; w.y &= 0xFFFF0000u; }
            if (!(two & 2u)) { w.z &= 0x0000FFFFu; w.w &= 0xFFFF0000u; }
          }
          uint32_t keyA, xA, yA, ciA, keyB, xB, yB, ciB;
          const bool okA = voxel_sample<FAST_DIV, BAND, SAFE, HASQ>(
              w.x, w.y, cA[j], p, q_min16, ibfe_off, ibfe_w, klo, khi, keyA, xA, yA, ciA, flags);
          const bool okB = voxel_sample<FAST_DIV, BAND, SAFE, HASQ>(
              w.z, w.w, cB[j], p, q_min16, ibfe_off, ibfe_w, klo, khi, keyB, xB, yB, ciB, flags);
          // tag: (round, wave) — two neighbouring queue reservations never share it
          const uint32_t tag = (((round & 15u) << 4) | wave_id()) << 24;
          if (fits) fits = voxel_pair_pass(L, tag, okA, keyA, xA, yA, ciA, okB, keyB, xB, yB, ciB);
        }
#ifdef RPL_ABL_WAITT
        {
          const unsigned long long t0 = clock64();
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          wt_latch += clock64() - t0;
        }
#endif
      }
#if defined(RPL_ABL_WAITT) || defined(RPL_ABL_LAT)
      if (p.dbg && threadIdx.x == 0) atomicAdd(&p.dbg[8 * b + 1], wt_latch);
      if (p.dbg && threadIdx.x == 64 * 7) atomicAdd(&p.dbg[8 * b + 2], wt_latch);
#endif
#endif  // RPL_ASM_RING
      if (!fits && lane_id() == 0) L.misc[2] = 1u;  // band does not fit
    };
    {
      using T_ = std::true_type;
      using F_ = std::false_type;
      if (first_band) {
        if (keepmask) stream(F_{}, T_{}, T_{});
        else if (q_min16) stream(F_{}, T_{}, F_{});
        else stream(F_{}, F_{}, F_{});
      } else if (keepmask) {
        stream(T_{}, T_{}, T_{});
      } else {
        stream(T_{}, T_{}, F_{});
      }
    }
    first_band = false;
    __syncthreads();
    RPL_MARK(0)

    auto bisect = [&]() {  // block-uniform: replace the band by its two halves
      if (threadIdx.x == 0) {
        uint32_t s = L.misc[3];
        if (klo == khi || s + 2 > 33) {
          L.misc[1] |= RPLGPU_SCAN_TABLE_FULL;  // cannot happen: one key is one cell/row
        } else {
          uint32_t mid = klo + (khi - klo) / 2;
          L.band_lo[s] = mid + 1;  // upper half is processed after the lower half
          L.band_hi[s] = khi;
          L.band_lo[s + 1] = klo;
          L.band_hi[s + 1] = mid;
          L.misc[3] = s + 2;
        }
      }
      __syncthreads();
    };
    if (L.misc[2]) {
      bisect();
      if (emit_mode == kEmitArenaFirst) emit_mode = kEmitCountOnly;
      continue;
    }

    // ---- phase R (out of line) ----------------------------------------------------------
    uint32_t ncell = 0;
    uint32_t out_limit = out_stride;  // cells this scan may write
    if (emit_mode == kEmitArenaKnown) {
      const unsigned long long room = arena_at >= arena.capacity ? 0ull : arena.capacity - arena_at;
      out_limit = (uint32_t)min(room, 0xFFFFFFFFull);
    } else if (emit_mode != kEmitLegacy) {
      out_limit = 0xFFFFFFFFu;  // (first band: bounded inside, at the reservation)
    }
    if (voxel_reduce(L, p, T.rcp, out, out_limit, b, &ncell, emit_mode, arena)) {
      bisect();
      if (emit_mode == kEmitArenaFirst) emit_mode = kEmitCountOnly;
      continue;
    }
    if (emit_mode == kEmitArenaFirst)  // (the reservation made inside voxel_reduce)
      arena_at = ((unsigned long long)L.tmp[29] << 32) | L.tmp[28];
    const uint32_t out_base = L.misc[7];
    __syncthreads();
    if (threadIdx.x == 0) L.misc[7] = out_base + ncell;
    __syncthreads();
    t_mark = clock64();
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
  }
  if (p.dbg && threadIdx.x == 0) atomicAdd(&p.dbg[8 * b], tacc[0]);
#undef RPL_MARK

  if (flags) atomicOr(&L.misc[1], flags);
  __syncthreads();
  if (threadIdx.x == 0) {
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
  if (threadIdx.x == 0) L.tmp[31] = gridDim.x + atomicAdd(&T.work_ctr[0], 1u);
  __syncthreads();  // LDS is reused by the next scan
  b = L.tmp[31];
  __syncthreads();
  }
#ifdef RPL_ABL_CLK
  if (p.dbg && threadIdx.x == 0) {  // developer aid: core clocks vs 100 MHz wall clock of this workgroup
    p.dbg[8 * blockIdx.x + 5] = clock64() - clk_c0;
    p.dbg[8 * blockIdx.x + 6] = wall_clock64() - clk_w0;
    p.dbg[8 * blockIdx.x + 7] = clk_w0;
    unsigned int xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.dbg[8 * blockIdx.x + 4] = ((unsigned long long)xcc << 32) | __smid();
  }
#endif
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

hipError_t launch_cloud_voxel(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, const uint32_t *keepmask, uint32_t mask_stride,
                              float *xyzi, uint32_t out_stride, uint32_t *n_points,
                              uint32_t *status, float *arena, unsigned long long arena_capacity,
                              unsigned long long *arena_cursor, unsigned long long *scan_start) {
  if (B == 0) return hipSuccess;
  VoxelArena ar;
  ar.base = (float4 *)arena;
  ar.cursor = arena_cursor;
  ar.capacity = arena_capacity;
  ar.scan_start = scan_start;
  // one persistent workgroup per CU of the handle's device; the scan queue is cleared by a
  // memset ahead of every launch (an aborted launch can therefore not poison the next one)
  uint32_t grid = std::min<uint32_t>(B, T.n_cu ? T.n_cu : 256u);
  if (const char *e = std::getenv("RPLGPU_VOXEL_GRID")) {  // developer aid
    const long g = std::atol(e);
    if (g > 0) grid = std::min<uint32_t>(B, (uint32_t)g);
  }
  if (hipError_t e = hipMemsetAsync(T.work_ctr, 0, 4, s); e != hipSuccess) return e;
#define RPL_LAUNCH_VOXEL(FD, SF)                                                              \
  hipLaunchKernelGGL((k_cloud_voxel<FD, SF>), dim3(grid), dim3(kBlock), 0, s, (const uint2 *)nodes, \
                     n_stride, n_per_scan, p, T, keepmask, mask_stride, (float4 *)xyzi, out_stride, \
                     n_points, status, B, ar)
  if (p.fast_div) {
    if (p.cell_range_safe) RPL_LAUNCH_VOXEL(true, true); else RPL_LAUNCH_VOXEL(true, false);
  } else {
    if (p.cell_range_safe) RPL_LAUNCH_VOXEL(false, true); else RPL_LAUNCH_VOXEL(false, false);
  }
#undef RPL_LAUNCH_VOXEL
  return hipGetLastError();
}

}  // namespace rpl
#ifdef RPL_ABL_NOGATHER
#undef cs
#endif
