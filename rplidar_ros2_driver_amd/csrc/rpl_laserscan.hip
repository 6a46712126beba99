// rpl_laserscan.hip — k_laserscan_a: the body of RPlidarNode::publish_scan in Mode A
// (scan_processing = true), reference src/rplidar_node.cpp:568-662, one 1024-thread
// workgroup per scan, one pass over HBM: 8 B read per sample, 8 B written per beam.
//
// What the reference does per scan                       what happens here
//   LOOP 1 (:583-602) keep dist != 0, convert units      interval test on dist_mm_q2 while the scan
//                                                        is loaded into registers (16 x dwordx4 per
//                                                        lane = 32 samples), block count -> beam_count
//   std::sort by angle_rad (:607-609)                    nothing: the winner of a bin is defined by a
//                                                        key, not by the visiting order (below)
//   index = (int)((angle - 0) / angle_increment) (:653)  angle from the host-built table (bit-exact by
//                                                        construction), divide = IEEE fp32 divide, or
//                                                        mul + 2 FMA when k_validate_idx proved it
//                                                        identical for EVERY (beam count, angle word)
//   if dist < ranges[index] take (strict <) (:657-660)   LDS ds_min_u64 on
//                                                          dist_m bits | angle word | intensity byte
//                                                        = first strict minimum in angle order; among
//                                                        samples with identical (dist_m, angle) the
//                                                        reference's winner is introsort's (unstable):
//                                                        ours is the smallest intensity, whatever the
//                                                        input order (deterministic, order-free)
//   ranges.assign(+inf), intensities.assign(0) (:640)    bins never hit flush as (+inf, 0)
//
// The input need not be sorted.  A window of 16 384 bins (128 KiB of LDS) is flushed with
// coalesced stores; scans with more beams take a second window over the cached bin indices
// (two u16 per VGPR), so no sample is converted twice.
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr uint32_t kLsWin = 16384;  // u64 bins per window
constexpr int kLsPairs = 16;        // dwordx4 (two samples) per lane: 16 * 1024 * 2 = 32768

typedef uint32_t ls_u32x4 __attribute__((ext_vector_type(4)));

// a / d as mul + exact FMA remainder + FMA correction with rd = RN(1/d) (Markstein); only
// used after k_validate_idx has compared it with the IEEE divide on this device for every
// operand pair that can occur.
__device__ __forceinline__ float ls_div(float a, float d, float rd) {
  float q = a * rd;
  float e = fmaf(-q, d, a);
  return fmaf(e, rd, q);
}

// angle_rad of a sample COMPUTED instead of gathered (round 6).  The reference's expressions
// (src/rplidar_node.cpp:588-589): angle_deg = q14 * 90.0f / 16384.0f is exact in fp32 (q14 * 45 has 22 bits,
// the rest is a power of two), angle_rad = (float)((double)angle_deg * (M_PI / 180.0f)) is one fp64 multiply
// by the same constant and one rounding to fp32 — IEEE operations the device has.  The wraps of :594-599
// never fire for a u16 input.  Like the cheap divides this is not taken on faith: k_validate_idx compares it
// with the host-built table for every angle word on this device, and only then (FAST) is it used.  It takes
// the 32 table gathers per lane out of the conversion step, which was bound by their issue rate (512 wave
// gathers per scan at ~30 cycles each: 15.4 k of the 65 k cycles a scan takes).  The inverted table
// (:646-651: one more fp64 subtraction, a compare and a second one near angle 0) stays a gather.
__device__ __forceinline__ float ls_angle(uint32_t q14) {
  const float deg = (float)q14 * 0.0054931640625f;  // q14 * 90.0f / 16384.0f, exact
  return (float)((double)deg * (M_PI / 180.0f));
}

// PAIRS: the sample pairs a lane works on.  Pair j of a lane holds samples 2 * (1024 j + lane)
// and the next one, so a scan of n samples only reaches the first ceil(n / 2048) pairs; the batch
// entry points use all 16, a single-scan call (its length is a launch argument) the instance that
// just covers it — for the scans a lidar really delivers (360 ... 8192 samples, one at a time)
// 16 waves converting 32 mostly all-zero samples per lane were most of the kernel: 21 -> 16 us per
// call at 360 samples, 24 -> 20 at 3200.
// MSG: the (single) scan goes straight into its serialised message (LsMsgOut, rpl_launch.hpp).
// ANG: angles computed (ls_angle) instead of gathered: FAST launches with the non-inverted table.
template <bool FAST, int PAIRS, bool MSG, bool ANG = false>
__global__ __launch_bounds__(kBlock) void k_laserscan_a(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, Tables T, const float *__restrict__ inc_table, const float *__restrict__ rinc_table,
    float *__restrict__ ranges, float *__restrict__ intens, uint32_t *__restrict__ beam_count,
    uint32_t n_given, LsMsgOut mo) {
  __shared__ unsigned long long s_bins[kLsWin];
  __shared__ uint32_t s_cnt;

  const uint32_t b = blockIdx.x;
  // (single scan through pinned host memory: the length comes as an argument — reading it from
  // the staging would be one more PCIe round trip in front of the node loads)
  const uint32_t n = min(n_given != 0xFFFFFFFFu ? n_given : n_per_scan[b], min(n_stride, kMaxN));
  const uint2 *scan = nodes + (size_t)b * n_stride;
  float *out_r = ranges + (size_t)b * n_stride;
  float *out_i = intens + (size_t)b * n_stride;
  (void)mo;

  if (threadIdx.x == 0) s_cnt = 0u;
#ifdef RPL_LS_DBG  // developer build: phase clocks of thread 0 (tools/dev/lsbench.py)
  unsigned long long dbg_t[7];
#define RPL_LS_CLK(i) dbg_t[i] = __builtin_amdgcn_s_memtime()
#else
#define RPL_LS_CLK(i) (void)0
#endif
  RPL_LS_CLK(0);

  // Bounds-checked buffer resource over the scan's n*8 bytes: what lies beyond reads as
  // zero = dist 0 = dropped by the keep test (d_lo >= 1).
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)scan, 0, (int)(n * 8u), 0x00020000);
  uint4 w[PAIRS];
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) {
    const ls_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(
        rsrc, (int)(((uint32_t)j * kBlock + threadIdx.x) * 16u), 0, 0);
    w[j] = make_uint4(t.x, t.y, t.z, t.w);
  }
  const uint32_t q_min16 = p.clip_enable ? (min(p.q_min, 256u) << 16) : 0u;
  auto keep = [&](uint32_t lo, uint32_t hi) -> bool {  // :584 (+ the optional E1 clip)
    const uint32_t d = __builtin_amdgcn_alignbit(hi, lo, 16);
    return ((d - p.d_lo) <= p.d_span) && ((hi & 0x00FF0000u) >= q_min16);
  };

  // LOOP 1: beam_count = number of kept samples (:634)
  uint32_t c = 0;
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) {
    c += keep(w[j].x, w[j].y) ? 1u : 0u;
    c += keep(w[j].z, w[j].w) ? 1u : 0u;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += (uint32_t)__shfl_xor((int)c, d, 64);
  __syncthreads();  // s_cnt = 0 visible
  if (lane_id() == 0 && c) atomicAdd(&s_cnt, c);
  __syncthreads();
  const uint32_t count = s_cnt;
  RPL_LS_CLK(1);  // loads + count
  if (threadIdx.x == 0) beam_count[b] = count;
  if (MSG && threadIdx.x == 0) *mo.msg_len = count ? mo.P.len + 8u * count + 4u : 0u;
  if (count == 0) return;  // :611-613 nothing published
  if (MSG) {
    // the message around the arrays (k_msg_laserscan's part): prefix template, then the words
    // that differ from scan to scan — stamp, the three count-dependent scalars with the
    // reference's expressions (:635-638: fp64 divides, one rounding to float), both array lengths
    uint32_t *msg = mo.msg;
    for (uint32_t i = threadIdx.x; i < mo.P.len / 4u; i += kBlock) msg[i] = mo.P.words[i];
    __syncthreads();  // the patches overwrite template words
    if (threadIdx.x == 0) {
      msg[mo.P.stamp_off / 4] = (uint32_t)mo.sec;
      msg[mo.P.stamp_off / 4 + 1] = mo.nanosec;
      const double denom = (double)count;
      float *f = reinterpret_cast<float *>(msg + mo.P.a_off / 4);
      f[2] = (float)((2.0 * M_PI) / denom);
      f[3] = (float)(mo.scan_duration / denom);
      f[4] = (float)mo.scan_duration;
      msg[mo.P.b_off / 4] = count;
      msg[mo.P.len / 4 + count] = count;  // intensities length word, right after ranges
    }
    out_r = reinterpret_cast<float *>(msg + mo.P.len / 4);
    out_i = out_r + count + 1u;
  }

  const float *lut = p.inverted ? T.angle_inv : T.angle;  // :588-599, :646-651
  // (float)(2*pi / (double)count), :635, and its reciprocal: the IEEE operations the host used
  // for inc_table / rinc_table (the values k_validate_idx checked), computed here instead of
  // fetched — a dependent memory round trip less behind the count
  const float inc = (float)(kTwoPi / (double)count);
  const float rinc = 1.0f / inc;
  (void)inc_table;
  (void)rinc_table;

  const uint32_t ishift = p.is_new_protocol ? 16u : 18u;  // :591-592 quality or quality >> 2
  const uint32_t imask = p.is_new_protocol ? 0xFFu : 0x3Fu;

  // Every sample becomes its 64-bit bin key IN PLACE (w[j] = {lowA, hiA, lowB, hiB}):
  //   hi = dist_m bits (:590), low = angle word << 16 | intensity byte << 8,
  // plus its bin index, two u16 per register (0xFFFF = not kept / guard :656).
  uint32_t idxp[PAIRS];
#pragma unroll
  for (int j = 0; j < PAIRS; ++j) {
    const uint4 ww = w[j];
    const float aA = ANG ? ls_angle(ww.x & 0xFFFFu) : lut[ww.x & 0xFFFFu];
    const float aB = ANG ? ls_angle(ww.z & 0xFFFFu) : lut[ww.z & 0xFFFFu];
    const float tA = FAST ? ls_div(aA, inc, rinc) : aA / inc;  // :653-654 (angle - 0.0f == angle)
    const float tB = FAST ? ls_div(aB, inc, rinc) : aB / inc;
    uint32_t iA = (uint32_t)(int)tA, iB = (uint32_t)(int)tB;
    if (!(keep(ww.x, ww.y) && iA < count)) iA = 0xFFFFu;
    if (!(keep(ww.z, ww.w) && iB < count)) iB = 0xFFFFu;
    idxp[j] = iA | (iB << 16);
    const float dfA = __uint2float_rn(__builtin_amdgcn_alignbit(ww.y, ww.x, 16));
    const float dfB = __uint2float_rn(__builtin_amdgcn_alignbit(ww.w, ww.z, 16));
    const float dmA = FAST ? ls_div(dfA, 4000.0f, 0.00025f) : dfA / 4000.0f;  // :590
    const float dmB = FAST ? ls_div(dfB, 4000.0f, 0.00025f) : dfB / 4000.0f;
    w[j] = make_uint4((ww.x << 16) | (((ww.y >> ishift) & imask) << 8), __float_as_uint(dmA),
                      (ww.z << 16) | (((ww.w >> ishift) & imask) << 8), __float_as_uint(dmB));
    if (j & 1) __builtin_amdgcn_sched_barrier(0);  // keep the conversion in place, 4 table loads in flight
  }
  RPL_LS_CLK(2);  // conversion
#ifdef RPL_LS_DBG
  dbg_t[3] = dbg_t[4] = dbg_t[5] = dbg_t[6] = dbg_t[2];
#endif
  auto key64 = [](uint32_t low, uint32_t hi) -> unsigned long long {
    return ((unsigned long long)hi << 32) | (unsigned long long)low;
  };

  for (uint32_t lo = 0; lo < count; lo += kLsWin) {
    const uint32_t nb = min(kLsWin, count - lo);
    for (uint32_t t = threadIdx.x; t < nb; t += kBlock) s_bins[t] = ~0ull;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PAIRS; ++j) {
      const uint32_t rA = (idxp[j] & 0xFFFFu) - lo, rB = (idxp[j] >> 16) - lo;
      if (rA < kLsWin) atomicMin(&s_bins[rA], key64(w[j].x, w[j].y));
      if (rB < kLsWin) atomicMin(&s_bins[rB], key64(w[j].z, w[j].w));
    }
    __syncthreads();
#ifdef RPL_LS_DBG
    if (lo == 0) RPL_LS_CLK(3); else RPL_LS_CLK(5);  // clear + atomics of a window
#endif
    for (uint32_t t = threadIdx.x; t < nb; t += kBlock) {
      const unsigned long long k = s_bins[t];
      const uint32_t hi = (uint32_t)(k >> 32), low = (uint32_t)k;
      const bool empty = hi == 0xFFFFFFFFu;  // no dist_m has this bit pattern (:640-641)
      // (streaming stores: the arrays are written once and read by a later kernel or the host)
      __builtin_nontemporal_store(__uint_as_float(empty ? 0x7F800000u : hi), &out_r[lo + t]);
      __builtin_nontemporal_store(empty ? 0.0f : (float)((low >> 8) & 0xFFu), &out_i[lo + t]);
    }
    __syncthreads();
#ifdef RPL_LS_DBG
    if (lo == 0) RPL_LS_CLK(4); else RPL_LS_CLK(6);  // flush of a window
#endif
  }
#ifdef RPL_LS_DBG
  if (threadIdx.x == 0 && n_stride >= count + 8u)  // (the unused tail of the scan's range row)
    for (int d = 0; d < 6; ++d) out_r[n_stride - 8 + d] = (float)(dbg_t[d + 1] - dbg_t[d]);
#endif
}

// ------------------------------------------------------------------------------
// k_validate_idx: for every beam count c <= max_count and every angle word (both tables),
// the bin index from the mul+2*FMA divide must equal the one from the IEEE divide.
// One block per beam count.
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_validate_idx(const float *__restrict__ angle,
                                                      const float *__restrict__ angle_inv,
                                                      const float *__restrict__ inc_table,
                                                      const float *__restrict__ rinc_table,
                                                      uint32_t *mismatches) {
  const uint32_t c = blockIdx.x + 1u;
  const float inc = inc_table[c], rinc = rinc_table[c];
  uint32_t bad = 0;
  for (uint32_t q = threadIdx.x; q < 65536u; q += 256u) {
    const float a0 = angle[q], a1 = angle_inv[q];
    if (c == 1u) bad += (__float_as_uint(ls_angle(q)) != __float_as_uint(a0));  // (once: block 0 checks ls_angle)
    bad += ((int)ls_div(a0, inc, rinc) != (int)(a0 / inc));
    bad += ((int)ls_div(a1, inc, rinc) != (int)(a1 / inc));
  }
  if (bad) atomicAdd(mismatches, bad);
}

hipError_t launch_validate_idx(hipStream_t s, const Tables &T, const float *inc_table,
                               const float *rinc_table, uint32_t max_count,
                               uint32_t *d_mismatches) {
  if (max_count == 0) return hipSuccess;
  hipLaunchKernelGGL(k_validate_idx, dim3(max_count), dim3(256), 0, s, T.angle, T.angle_inv,
                     inc_table, rinc_table, d_mismatches);
  return hipGetLastError();
}

hipError_t launch_laserscan_a(hipStream_t s, const void *nodes, uint32_t n_stride,
                              const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                              const Tables &T, const float *inc_table, const float *rinc_table,
                              bool fast, float *ranges, float *intens, uint32_t *beam_count,
                              uint32_t n_given, const LsMsgOut *msg_out) {
  if (B == 0) return hipSuccess;
  if (B != 1) n_given = 0xFFFFFFFFu;
  // the instance that just covers a single scan of known length (see the kernel's head)
  const uint32_t pairs_needed = n_given == 0xFFFFFFFFu ? (uint32_t)kLsPairs
                                                       : (std::min(n_given, kMaxN) + 2u * kBlock - 1u) / (2u * kBlock);
  // (the message form exists for single scans on the validated fast path; the caller checks)
  const bool to_msg = msg_out != nullptr;
  if (to_msg && !(B == 1u && fast)) return hipErrorInvalidValue;
  const LsMsgOut mo = to_msg ? *msg_out : LsMsgOut{};
  const bool ang = fast && !p.inverted;  // (see ls_angle)
#define RPL_LAUNCH_LS_A(F, P, M, A)                                                                 \
  hipLaunchKernelGGL((k_laserscan_a<F, P, M, A>), dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, \
                     n_stride, n_per_scan, p, T, inc_table, rinc_table, ranges, intens, beam_count,  \
                     n_given, mo)
#define RPL_LAUNCH_LS(F, P)                                                                         \
  do {                                                                                              \
    if (F && to_msg) {                                                                              \
      if (ang) RPL_LAUNCH_LS_A(F, P, F, F); else RPL_LAUNCH_LS_A(F, P, F, false);                   \
    } else {                                                                                        \
      if (F && ang) RPL_LAUNCH_LS_A(F, P, false, F); else RPL_LAUNCH_LS_A(F, P, false, false);      \
    }                                                                                               \
  } while (0)
#define RPL_LAUNCH_LS_P(F)                      \
  do {                                          \
    if (pairs_needed <= 1u) RPL_LAUNCH_LS(F, 1);       \
    else if (pairs_needed <= 2u) RPL_LAUNCH_LS(F, 2);  \
    else if (pairs_needed <= 4u) RPL_LAUNCH_LS(F, 4);  \
    else if (pairs_needed <= 8u) RPL_LAUNCH_LS(F, 8);  \
    else RPL_LAUNCH_LS(F, kLsPairs);            \
  } while (0)
  if (fast) RPL_LAUNCH_LS_P(true); else RPL_LAUNCH_LS_P(false);
#undef RPL_LAUNCH_LS_P
#undef RPL_LAUNCH_LS
#undef RPL_LAUNCH_LS_A
  return hipGetLastError();
}

}  // namespace rpl
