// rpl_ror.hip — k_ror_mask: radius-outlier-removal keep mask (extension E5 of SURVEY.md §8
// a-ext): a kept point survives iff at least `k` OTHER kept points of the same scan lie
// within `r` (fp32: dx*dx + dy*dy <= r*r, products then sum, no FMA — the oracle's
// orc_ror_mask expression, oracle/oracle.cpp).  One 1024-thread workgroup owns one scan.
//
// The oracle is O(n^2).  A point only has to be shown to have k neighbours, or to have fewer
// after every candidate was seen, so the kernel works in two stages:
//   1. index neighbours.  A lidar scan is ordered by angle: the nearest points of sample i are
//      almost always samples i+-1, i+-2, ...  Every kept sample tests its 8 index neighbours
//      (coalesced loads of the shifted scan) and is settled as soon as k of them lie within r.
//   1b. the samples that stay unsettled are mostly islands between drop-outs whose spatial
//      neighbours are a few more indices away: ONE WAVE per unsettled sample tests the 64 samples
//      before and the 64 after it (two ballots).  Only what is still unsettled then goes on;
//      on ring-like scans that is nothing, and the row structure of stage 2 (two more passes
//      over the whole scan) is not built at all (C5 at 4096 scans: 2.56 -> see DESIGN.md).
//   2. the few samples that stay unsettled (isolated returns, real outliers) are searched
//      exhaustively but cooperatively: the kept samples are binned by y
//      into rows of height h >= 1.001 r (h grows with the scan's y extent so that 2048 rows
//      always suffice) with a counting sort in LDS, and ONE WAVE per unsettled sample sweeps
//      the candidates of its own row and the two rows next to it, 64 at a time (ballot +
//      popcount), stopping at k.
// Candidate coordinates are always recomputed from the raw node and the (cos, sin) table,
// exactly as the cloud kernels compute them, so the distance test sees bit-identical operands
// and the mask is bit-exact.  (The first version gave every thread its 32 samples and walked the
// three-row slice alone, two dependent gathers per candidate: 9 ms per scan.)
//
// Output: one bit per input sample (bit i of word i/32), 1 = the sample passed E1 AND E5.
#include <algorithm>

#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

constexpr uint32_t kRorRows = 2048;
constexpr uint32_t kRorTodo = 8192;  // unsettled samples stage 2 can take (else: next round)
constexpr int kRorNear = 4;          // stage 1 looks at samples i-4 .. i+4

constexpr int kRorPer = 2;           // stage 1: samples per thread and trip (4: measured slower, 46 k against 42 k cycles)
constexpr uint32_t kRorSpan = (uint32_t)kRorPer * kBlock;
constexpr uint32_t kRorWin = kRorSpan + 2u * kRorNear;  // a trip's samples + halo

struct RorLds {
  // stage 2: sample indices grouped by row (uint16_t[kMaxN]); stage 1 (over by then): two windows
  // of (x, y) — the samples of a trip + halo; two of them, so that a trip needs ONE barrier (a
  // wave may publish trip t + 1 while another still reads trip t)
  alignas(16) unsigned char idx_or_win[2 * kRorWin * 8 > kMaxN * 2 ? 2 * kRorWin * 8 : kMaxN * 2];
  uint32_t rowstart[kRorRows];  // first slot of a row
  uint32_t rowfill[kRorRows];   // one past its last slot (after the scatter)
  uint32_t misc[8];             // 0/1 ymin/ymax (order-preserving uint encoding), 2 #unsettled
  uint32_t tmp[32];
  uint16_t todo[kRorTodo];      // unsettled samples of stage 1
  uint16_t todo2[kRorTodo];     // ... still unsettled after the +-64 window of stage 1b
  uint32_t late[kMaxN / 32];    // keep bits found by stage 2 (bit i of word i/32)
};

// order-preserving float <-> uint map (for LDS atomicMin / atomicMax on floats)
__device__ __forceinline__ uint32_t f2ord(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u);
}

__device__ __forceinline__ float2 node_xy(uint2 nd, const float2 *__restrict__ cs) {
  const float dm = nd_dist_m(nd_dist(nd));  // :590
  const float2 c = cs[nd_q14(nd)];
  return make_float2(dm * c.x, dm * c.y);  // E2
}

// FAST: dist / 4000 as mul + 2 FMA (validated bit-identical to the IEEE divide on this device,
// rplgpu_api.hip validate_divisor) in stage 1, where every sample is converted
// LISTED (round 6): the scans of the work items the voxel kernel's ROR instance could not settle itself
// (T.redo: count word, then item numbers; an item = `group` consecutive scans of the `n_scans`), a
// persistent grid that learns the count on the device — usually zero, and the launch ends at once.
template <bool FAST, bool LISTED>
__global__ __launch_bounds__(kBlock) void k_ror_mask(const uint2 *__restrict__ nodes,
                                                     uint32_t n_stride,
                                                     const uint32_t *__restrict__ n_per_scan,
                                                     KParams p, Tables T,
                                                     uint32_t *__restrict__ mask_out,
                                                     uint32_t mask_stride, uint32_t group,
                                                     uint32_t n_scans) {
  __shared__ RorLds L;
  uint16_t *const L_idx = reinterpret_cast<uint16_t *>(L.idx_or_win);
  float2 *const L_win = reinterpret_cast<float2 *>(L.idx_or_win);
  for (uint32_t item = blockIdx.x;; item += gridDim.x) {  // (not LISTED: one trip, the scan blockIdx.x)
  uint32_t b = item;
  if (LISTED) {
    const uint32_t listed = (uint32_t)__builtin_amdgcn_readfirstlane((int)T.redo[0]);
    if (item / group >= listed) break;
    b = T.redo[4u + item / group] * group + item % group;
    if (b >= n_scans) continue;  // (the last group of a batch may be short)
  } else if (item != blockIdx.x) {
    break;
  }
#ifdef RPL_ROR_DBG
  const unsigned long long dbg_entry = __builtin_amdgcn_s_memtime();
#endif
  const uint32_t n = min(n_per_scan[b], min(n_stride, kMaxN));  // never past the slot
  const uint2 *scan = nodes + (size_t)b * n_stride;
  uint32_t *mask = mask_out + (size_t)b * mask_stride;
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
  const uint32_t wave = wave_id(), lane = lane_id();

  if (threadIdx.x == 0) {
    L.misc[0] = 0xFFFFFFFFu;
    L.misc[1] = 0u;
    L.misc[2] = 0u;
    L.misc[3] = 0u;
  }
  for (uint32_t t = threadIdx.x; t < kRorRows; t += kBlock) L.rowstart[t] = 0u;
  for (uint32_t t = threadIdx.x; t < kMaxN / 32u; t += kBlock) L.late[t] = 0u;
  __syncthreads();

#ifdef RPL_ROR_DBG  // developer build: phase clocks of thread 0 into the cycle buffer (tools/dev/rorbench.py)
  unsigned long long dbg_t[6];
#define RPL_ROR_CLK(i) dbg_t[i] = __builtin_amdgcn_s_memtime()
#else
#define RPL_ROR_CLK(i) (void)0
#endif
  RPL_ROR_CLK(0);
  const float r2 = p.ror_r2;
  const uint32_t need = p.ror_k;
  // ---- stage 1: E1 keep bits, y extent, and the index-neighbour test ----------------------
  uint32_t kept = 0, keep = 0;
  // A trip covers the 2048 consecutive samples [2048 t, 2048 t + 2048): every thread computes
  // the points of its two samples ONCE and publishes them in LDS (a sample that is not kept, or
  // lies outside the scan, as NaN: it then fails every distance test), four halo samples on
  // either side come from eight extra threads; the index neighbours are read from there, and the
  // raw nodes of the next trip are already on their way.  (Every thread loading and converting
  // its eight neighbours itself — nine node loads and nine table gathers per sample behind
  // data-dependent branches — made this stage 1.9 ms of C5's 4.3; one 1024-sample window per
  // trip without prefetch 1.35.)
  // A sample that is not kept (or lies outside the scan) is published as a point so far away
  // that its squared distance from any real point overflows to +inf: then r2 - d2 = -inf, and the
  // whole test "d2 <= r2" is the SIGN BIT of r2 - d2 (0 = within; equal gives +0; no NaN can
  // arise: real coordinates are below 1.1e6 m).  The eight sign bits are shifted into one word
  // (v_alignbit) and counted once — a compare, a select and the bit assembly per neighbour less.
  const float far = 1.0e30f;
  auto load_node = [&](uint32_t q) -> uint2 { return q < n ? scan[q] : make_uint2(0u, 0u); };  // dist 0: dropped
  constexpr uint32_t kSpan = kRorSpan;
  const int trips = (int)((n + kSpan - 1u) / kSpan);  // block-uniform
  // software pipeline: raw nodes two trips ahead, the table gathers one trip ahead; the eight halo
  // samples are one more slot of threads 0..7 and ride the same pipeline
  // (thread 3's offset base - 1 is 0xFFFFFFFF as a number: "no halo duty" is a flag of its own,
  // not a sentinel value of the offset)
  const bool has_halo = threadIdx.x < 2u * kRorNear;
  const bool halo_wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) == 0;  // (scalar: the eight halo lanes live in wave 0)
  const uint32_t halo_q = threadIdx.x < (uint32_t)kRorNear ? threadIdx.x - (uint32_t)kRorNear  // base - 4 + t
                                                           : kSpan + threadIdx.x - (uint32_t)kRorNear;
  auto halo_index = [&](uint32_t base) -> uint32_t {
    // (first trip: base - 4 + t wraps below 0 and fails q < n; no duty: always out of range)
    return has_halo ? base + halo_q : 0xFFFFFFFFu;
  };
  // gathered (cos, sin) of a node: always a valid table index, whatever the node holds
  auto cs_of = [&](uint2 c) -> float2 { return cs[nd_q14(c)]; };
  // the point of a node from its gathered (cos, sin): node_xy's arithmetic (E2), `far` if not kept
  auto point_from = [&](uint2 c, float2 g) -> float2 {
    const float df = __uint2float_rn(nd_dist(c));
    float dm;  // :590
    if (FAST) {
      const float q = df * 0.00025f, e = fmaf(-q, 4000.0f, df);
      dm = fmaf(e, 0.00025f, q);
    } else {
      dm = df / 4000.0f;
    }
    const bool k = nd_keep(nd_dist(c), nd_quality(c), p);
    return make_float2(k ? dm * g.x : far, k ? dm * g.y : far);
  };
  // nodes of trip 0 and their table entries, nodes of trip 1
  uint2 c[kRorPer], nx[kRorPer], ch = load_node(halo_index(0u));
  float2 g[kRorPer], gh = cs_of(ch);
#pragma unroll
  for (int h = 0; h < kRorPer; ++h) {
    c[h] = load_node((uint32_t)h * kBlock + threadIdx.x);
    g[h] = cs_of(c[h]);
  }
#pragma unroll
  for (int h = 0; h < kRorPer; ++h) nx[h] = load_node(kSpan + (uint32_t)h * kBlock + threadIdx.x);
  uint2 nh = load_node(halo_index(kSpan));
  for (int t = 0; t < trips; ++t) {
    const uint32_t base = (uint32_t)t * kSpan;
    // Vector loads return in order.  The table gathers of a trip are issued back to back a trip
    // ahead, IN FRONT of the node loads of the trip after it, and are only consumed here: the
    // wait for them leaves the younger node loads in flight.  (With the gather inside the
    // kept-test branch of every point, each was followed by a wait for everything in flight —
    // three L2 round trips in a row per trip and, behind the node loads, an HBM one: stage 1 was
    // 103 k cycles per scan.)
    float2 me[kRorPer];
#pragma unroll
    for (int h = 0; h < kRorPer; ++h) me[h] = point_from(c[h], g[h]);
    float2 meh = make_float2(far, far);
    if (halo_wave) meh = point_from(ch, gh);  // (the eight halo lanes live in wave 0)
#pragma unroll
    for (int h = 0; h < kRorPer; ++h) {
      c[h] = nx[h];  // nodes of trip t + 1 (loaded a trip ago)
      g[h] = cs_of(c[h]);
    }
    if (halo_wave) {
      ch = nh;
      gh = cs_of(ch);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int h = 0; h < kRorPer; ++h)  // nodes of trip t + 2
      nx[h] = load_node(base + 2u * kSpan + (uint32_t)h * kBlock + threadIdx.x);
    if (halo_wave) nh = load_node(halo_index(base + 2u * kSpan));
    __builtin_amdgcn_sched_barrier(0);
    float2 *win = L_win + (size_t)(t & 1) * kRorWin;
#pragma unroll
    for (int h = 0; h < kRorPer; ++h) win[kRorNear + h * kBlock + threadIdx.x] = me[h];
    if (threadIdx.x < (uint32_t)kRorNear) win[threadIdx.x] = meh;
    else if (threadIdx.x < 2u * kRorNear) win[kSpan + threadIdx.x] = meh;
    __syncthreads();
#pragma unroll
    for (int h = 0; h < kRorPer; ++h) {
      const float2 m = me[h];
      const uint32_t i = base + (uint32_t)h * kBlock + threadIdx.x;
      const int j = kRorPer * t + h;  // bit j of kept / keep: sample 1024 j + thread
      const bool is_kept = m.x < 1.0e29f;
      // the neighbours at +-1, +-2 first; the outer four only if some lane of the wave is still
      // short (on ring-like scans with 10 % drop-outs: one wave pass in five)
      uint32_t outside = 0;  // one bit per tested neighbour: d2 > r2
      const int at = (int)(kRorNear + threadIdx.x) + h * kBlock;
      auto test_ring = [&](int o) {
#pragma unroll
        for (int sgn = 0; sgn < 2; ++sgn) {
          const float2 pc = win[at + (sgn ? -o : o)];
          const float dx = m.x - pc.x, dy = m.y - pc.y;
          const float d2 = dx * dx + dy * dy;  // products then sum (-ffp-contract=off)
          outside = __builtin_amdgcn_alignbit(outside, __float_as_uint(r2 - d2), 31);
        }
      };
      static_assert(kRorNear == 4, "two rounds of two rings");
      test_ring(1);
      test_ring(2);
      uint32_t cnt = 4u - (uint32_t)__builtin_popcount(outside);
      if (__builtin_amdgcn_ballot_w64(is_kept && cnt < need) != 0ull) {  // (wave-uniform)
        test_ring(3);
        test_ring(4);
        cnt = 8u - (uint32_t)__builtin_popcount(outside);
      }
      if (is_kept) {
        kept |= 1u << j;
        if (cnt >= need) {
          keep |= 1u << j;
        } else {  // unsettled: stage 1b / 2
          const uint32_t slot = atomicAdd(&L.misc[2], 1u);
          if (slot < kRorTodo) L.todo[slot] = (uint16_t)i;
        }
      }
    }
  }
  __syncthreads();
  RPL_ROR_CLK(1);
  uint32_t n_todo1 = L.misc[2];  // block-uniform
  // ---- a short scan (A1-class: 360 ... 1024 samples) still has ALL its points in the stage-1
  // window: every unsettled sample is
  // settled right here, exhaustively and from LDS — a wave per sample, 64 candidates at a time —
  // instead of going through stages 1b and 2 (a 360-sample scan, where nearly every point is
  // farther than r from its index neighbours, spent ~100 k cycles building the row structure).
  // (only for really short scans, <= 1024 samples: a true outlier walks ALL blocks of 64
  // candidates, ~120 cycles each; at 3200 samples the stages below — two block steps per sample in
  // 1b, then the handful of leftovers against every thread's samples — are the cheaper way: 52
  // against 68 us per call)
  if (n <= 1024u && n_todo1 != 0u && n_todo1 <= kRorTodo) {
    auto point = [&](uint32_t q) -> float2 {
      return L_win[(q < kSpan ? 0u : kRorWin) + kRorNear + (q & (kSpan - 1u))];
    };
    static_assert((kRorSpan & (kRorSpan - 1u)) == 0u, "the window index is a mask");
    const uint32_t wave_u = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave);
    for (uint32_t t = wave_u; t < n_todo1; t += kWaves) {
      const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.todo[t]);  // (scalar)
      const float2 me = point(i);
      uint32_t cnt = 0;
      // (blocks of 64 candidates, the sample's own block first and then outwards on both sides:
      // what can be settled at all is settled within a step or two)
      const uint32_t nblk = (n + 63u) >> 6, bi = i >> 6;
      auto hits_in = [&](uint32_t blk) -> uint32_t {
        const uint32_t q = blk * 64u + lane;
        bool hit = false;
        if (q < n && q != i) {
          const float2 pc = point(q);  // (a sample that is not kept sits 1e30 m away)
          const float dx = me.x - pc.x, dy = me.y - pc.y;
          const float d2 = dx * dx + dy * dy;
          hit = d2 <= r2;
        }
        return (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
      };
      cnt = hits_in(bi);
      for (uint32_t d = 1; d < nblk && cnt < need; ++d) {
        if (bi + d < nblk) cnt += hits_in(bi + d);
        if (d <= bi) cnt += hits_in(bi - d);
      }
      if (lane == 0 && cnt >= need) atomicOr(&L.late[i >> 5], 1u << (i & 31u));
    }
    __syncthreads();
    n_todo1 = 0u;  // (settled: nothing goes on to the stages below)
  }
  // ---- stage 1b: a wave per unsettled sample, the 64 samples before and after it -------------
  if (n_todo1 != 0u && n_todo1 <= kRorTodo) {
    for (uint32_t t = wave; t < n_todo1; t += kWaves) {
      const uint32_t i = L.todo[t];
      const float2 me = node_xy(scan[i], cs);  // wave-uniform address
      uint32_t cnt = 0;
#pragma unroll
      for (int side = 0; side < 2; ++side) {
        const uint32_t q = side ? i + 1u + lane : i - 1u - lane;  // (wraps below 0: fails q < n)
        bool hit = false;
        if (q < n) {
          const uint2 c = scan[q];
          if (nd_keep(nd_dist(c), nd_quality(c), p)) {
            const float2 pc = node_xy(c, cs);
            const float dx = me.x - pc.x, dy = me.y - pc.y;
            const float d2 = dx * dx + dy * dy;
            hit = d2 <= r2;
          }
        }
        cnt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
      }
      if (lane == 0) {
        if (cnt >= need) {
          atomicOr(&L.late[i >> 5], 1u << (i & 31u));
        } else {
          L.todo2[atomicAdd(&L.misc[3], 1u)] = (uint16_t)i;
        }
      }
    }
    __syncthreads();
  }
  RPL_ROR_CLK(2);
  // (a scan with more unsettled samples than the list holds skips 1b: stage 2 re-examines all)
  const uint32_t n_todo_all = n_todo1 <= kRorTodo ? L.misc[3] : n_todo1;  // block-uniform
  const uint16_t *todo = n_todo1 <= kRorTodo ? L.todo2 : L.todo;
  // A handful of leftovers (the usual case when there are any: isolated returns): every thread
  // runs its own kept samples past them, instead of building the row structure for the whole scan
  // (two passes and a sort: ~100 k cycles for what is typically one or two samples).
  constexpr uint32_t kRorFew = 4;
  if (n_todo_all != 0u && n_todo_all <= kRorFew) {
    float2 *few = L_win;           // (stage 1 is over: its window is free)
    uint32_t *few_cnt = L.tmp;     // hits per leftover
    if (threadIdx.x < n_todo_all) {
      few[threadIdx.x] = node_xy(scan[todo[threadIdx.x]], cs);
      few_cnt[threadIdx.x] = 0u;
    }
    __syncthreads();
    uint32_t hits[kRorFew] = {0u, 0u, 0u, 0u};
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        const float2 pc = node_xy(scan[i], cs);
#pragma unroll
        for (uint32_t u = 0; u < kRorFew; ++u) {
          if (u < n_todo_all) {  // (block-uniform)
            const float2 me = few[u];
            const float dx = me.x - pc.x, dy = me.y - pc.y;
            const float d2 = dx * dx + dy * dy;
            hits[u] += (i != (uint32_t)todo[u] && d2 <= r2) ? 1u : 0u;
          }
        }
      }
    }
#pragma unroll
    for (uint32_t u = 0; u < kRorFew; ++u) {
      if (u < n_todo_all) {
        const uint32_t tot = wave_incl_scan_fast(hits[u]);  // (the wave's sum in lane 63)
        if (lane == 63u && tot) atomicAdd(&few_cnt[u], tot);
      }
    }
    __syncthreads();
    if (threadIdx.x < n_todo_all && few_cnt[threadIdx.x] >= need) {
      const uint32_t i = todo[threadIdx.x];
      atomicOr(&L.late[i >> 5], 1u << (i & 31u));
    }
  } else if (n_todo_all != 0u) {
    // the y extent of the kept samples (only this path needs it: one more pass over the scan)
    {
      float ymin = __uint_as_float(0x7F800000u), ymax = __uint_as_float(0xFF800000u);
      for (int j = 0; j < kIters; ++j) {
        if ((kept >> j) & 1u) {
          const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
          const float y = node_xy(scan[i], cs).y;
          ymin = fminf(ymin, y);
          ymax = fmaxf(ymax, y);
        }
      }
#pragma unroll
      for (int d = 32; d > 0; d >>= 1) {
        ymin = fminf(ymin, __shfl_xor(ymin, d, 64));
        ymax = fmaxf(ymax, __shfl_xor(ymax, d, 64));
      }
      if (lane == 0 && ymin <= ymax) {
        atomicMin(&L.misc[0], f2ord(ymin));
        atomicMax(&L.misc[1], f2ord(ymax));
      }
      __syncthreads();
    }
    const float y0 = ord2f(L.misc[0]), y1 = ord2f(L.misc[1]);
    const float r = sqrtf(p.ror_r2);
    // rows at least 1.001 r high (points within r in y are at most one row apart) and few
    // enough to fit the table whatever the extent
    const float h = fmaxf(r * 1.001f, (y1 - y0) / (float)(kRorRows - 2));
    const float inv_h = 1.0f / h;
    auto row_of = [&](float y) -> uint32_t {
      const float t = (y - y0) * inv_h;
      return min((uint32_t)fmaxf(t, 0.0f), kRorRows - 1u);
    };
    // ---- counting sort of the kept sample indices by row -------------------------------
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        atomicAdd(&L.rowstart[row_of(node_xy(scan[i], cs).y)], 1u);
      }
    }
    __syncthreads();
    {
      const uint32_t r0 = L.rowstart[2 * threadIdx.x], r1 = L.rowstart[2 * threadIdx.x + 1];
      uint32_t tot;
      const uint32_t ex = block_excl_scan(r0 + r1, L.tmp, &tot);
      L.rowstart[2 * threadIdx.x] = ex;
      L.rowstart[2 * threadIdx.x + 1] = ex + r0;
      L.rowfill[2 * threadIdx.x] = ex;
      L.rowfill[2 * threadIdx.x + 1] = ex + r0;
    }
    __syncthreads();
    for (int j = 0; j < kIters; ++j) {
      if ((kept >> j) & 1u) {
        const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
        const uint32_t pos = atomicAdd(&L.rowfill[row_of(node_xy(scan[i], cs).y)], 1u);
        L_idx[pos] = (uint16_t)i;
      }
    }
    __syncthreads();
    RPL_ROR_CLK(3);
    // ---- stage 2: one wave per unsettled sample, 64 candidates at a time ---------------------
    const uint32_t n_todo = min(n_todo_all, kRorTodo);
    for (uint32_t t = wave; t < n_todo; t += kWaves) {
      const uint32_t i = todo[t];
      const float2 me = node_xy(scan[i], cs);  // wave-uniform address
      const uint32_t row = row_of(me.y);
      const uint32_t a = L.rowstart[row > 0u ? row - 1u : 0u];
      const uint32_t e = L.rowfill[min(row + 1u, kRorRows - 1u)];
      uint32_t cnt = 0;
      for (uint32_t q0 = a; q0 < e && cnt < need; q0 += 64u) {
        const uint32_t q = q0 + lane;
        bool hit = false;
        if (q < e) {
          const uint32_t jc = L_idx[q];
          const float2 pc = node_xy(scan[jc], cs);
          const float dx = me.x - pc.x, dy = me.y - pc.y;
          const float d2 = dx * dx + dy * dy;
          hit = (jc != i) && (d2 <= r2);
        }
        cnt += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(hit));
      }
      if (lane == 0 && cnt >= need) atomicOr(&L.late[i >> 5], 1u << (i & 31u));
    }
    // more unsettled samples than the list holds (an adversarial scan): the rest is done the
    // slow way, one thread per sample over the same three-row slice
    if (n_todo_all > kRorTodo) {
      __syncthreads();
      // which samples are in the list is arbitrary, so every unsettled sample not yet kept is
      // simply re-examined (idempotent)
      for (int j = 0; j < kIters; ++j) {
        if (((kept & ~keep) >> j) & 1u) {
          const uint32_t i = ((uint32_t)(j * kWaves) + wave) * 64u + lane;
          if ((L.late[i >> 5] >> (i & 31u)) & 1u) continue;
          const float2 me = node_xy(scan[i], cs);
          const uint32_t row = row_of(me.y);
          const uint32_t a = L.rowstart[row > 0u ? row - 1u : 0u];
          const uint32_t e = L.rowfill[min(row + 1u, kRorRows - 1u)];
          uint32_t cnt = 0;
          for (uint32_t q = a; q < e && cnt < need; ++q) {
            const uint32_t jc = L_idx[q];
            const float2 pc = node_xy(scan[jc], cs);
            const float dx = me.x - pc.x, dy = me.y - pc.y;
            const float d2 = dx * dx + dy * dy;
            cnt += (jc != i && d2 <= r2) ? 1u : 0u;
          }
          if (cnt >= need) keep |= 1u << j;
        }
      }
    }
  }
  __syncthreads();
  RPL_ROR_CLK(4);
#ifdef RPL_ROR_DBG
  if (n_todo_all == 0u) dbg_t[3] = dbg_t[2];
  if (threadIdx.x == 0 && p.dbg) {
    for (int d = 0; d < 4; ++d) p.dbg[(size_t)b * 16 + d] = dbg_t[d + 1] - dbg_t[d];
    p.dbg[(size_t)b * 16 + 4] = n_todo1;
    p.dbg[(size_t)b * 16 + 5] = n_todo_all;
  }
#endif
  // ---- one bit per sample: chunk c = j*16 + wave covers samples [64c, 64c+64) -------------
  // (the stage-1 bits join the late bits in LDS — every lane of a wave ends up with one of the
  // wave's 64 mask words — and the row of mask words leaves in one coalesced store)
  static_assert(kIters == 32, "one mask word per lane: lanes 2j and 2j+1 take the ballot of bit j");
  uint32_t mine = 0;
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const uint64_t m = __ballot((keep >> j) & 1u);
    if ((int)(lane >> 1) == j) mine = (lane & 1u) ? (uint32_t)(m >> 32) : (uint32_t)m;
  }
  atomicOr(&L.late[2u * ((lane >> 1) * (uint32_t)kWaves + wave) + (lane & 1u)], mine);
  __syncthreads();
  for (uint32_t t = threadIdx.x; t < kMaxN / 32u && t < mask_stride; t += kBlock) mask[t] = L.late[t];
  if (LISTED) __syncthreads();  // (the next listed scan reuses the LDS)
#ifdef RPL_ROR_DBG
  __syncthreads();
  if (threadIdx.x == 0 && p.dbg) {
    p.dbg[(size_t)b * 16 + 6] = dbg_t[0] - dbg_entry;                          // prologue
    p.dbg[(size_t)b * 16 + 7] = __builtin_amdgcn_s_memtime() - dbg_t[4];       // mask write-out
    p.dbg[(size_t)b * 16 + 8] = dbg_entry;                                     // (absolute: gaps between workgroups)
    p.dbg[(size_t)b * 16 + 9] = __builtin_amdgcn_s_memtime();
    p.dbg[(size_t)b * 16 + 10] = __builtin_amdgcn_s_getreg(((6 - 1) << 11) | (0 << 6) | 4) ;  // HW_ID
  }
#endif
  }
}

hipError_t launch_ror_mask(hipStream_t s, const void *nodes, uint32_t n_stride,
                           const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                           const Tables &T, uint32_t *mask, uint32_t mask_stride, bool listed,
                           uint32_t group) {
  if (B == 0) return hipSuccess;
  if (listed) {
    if (!T.redo || !p.fast_d4000 || group == 0) return hipErrorInvalidValue;
    const uint32_t grid = std::min<uint32_t>(B, T.n_cu ? T.n_cu : 256u);
    hipLaunchKernelGGL((k_ror_mask<true, true>), dim3(grid), dim3(kBlock), 0, s, (const uint2 *)nodes,
                       n_stride, n_per_scan, p, T, mask, mask_stride, group, B);
    return hipGetLastError();
  }
  if (p.fast_d4000)
    hipLaunchKernelGGL((k_ror_mask<true, false>), dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                       n_per_scan, p, T, mask, mask_stride, 1u, B);
  else
    hipLaunchKernelGGL((k_ror_mask<false, false>), dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                       n_per_scan, p, T, mask, mask_stride, 1u, B);
  return hipGetLastError();
}

}  // namespace rpl
