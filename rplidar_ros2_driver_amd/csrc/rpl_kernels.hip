// rpl_kernels.hip — hand-written gfx950 (CDNA4) kernels for the RPLIDAR scan path.
//
// No MFMA anywhere: this is filtering / binning / compaction (HBM-bound integer and
// fp32 work).  What matters is coalesced 8-byte node streaming, wave64 ballots and
// prefix sums for the keep mask, and LDS-staged per-bin / per-cell reductions.
//
// Reference semantics reproduced (citations relative to /root/reference):
//   k_ascend      src/sdk/src/sl_lidar_driver.cpp:128-184  (ascendScanData_)
//   k_laserscan   src/rplidar_node.cpp:568-680             (publish_scan body)
//   k_cloud*      extensions E1..E4 of SURVEY.md §8(a-ext)
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math): every
// fp32 divide below must be the correctly rounded IEEE divide and no mul+add may be
// fused, otherwise bin / cell indices stop being bit-exact.
#include "rpl_device.hpp"
#include "rpl_launch.hpp"

namespace rpl {

// ------------------------------------------------------------------------------
// Resident-scan helpers
// ------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sample_index(int j) {
  return ((uint32_t)(j * kWaves) + wave_id()) * 64u + lane_id();
}

// Bounds-checked buffer loads over the scan's n*8 bytes: a sample beyond the scan reads as
// zero (dist 0 = invalid), every lane always issues the load (no exec branches, no saved
// addresses).
typedef uint32_t k_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_scan(const uint2 *__restrict__ scan, uint32_t n,
                                          uint2 (&v)[kIters]) {
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)scan, 0, (int)(n * 8u), 0x00020000);
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    const k_u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)(sample_index(j) * 8u), 0, 0);
    v[j] = make_uint2(t.x, t.y);
  }
}

// Publishes one ballot per chunk, then turns the 512 chunk populations into
// exclusive bases.  Returns the total.  All 1024 threads must call.
__device__ __forceinline__ uint32_t publish_masks_and_scan(uint32_t flagbits, uint64_t *s_mask,
                                                           uint32_t *s_cbase, uint32_t *s_tmp) {
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    uint64_t m = __ballot((flagbits >> j) & 1u);
    if (lane_id() == 0) s_mask[j * kWaves + wave_id()] = m;
  }
  __syncthreads();
  uint32_t cnt = (threadIdx.x < kChunks) ? (uint32_t)__popcll(s_mask[threadIdx.x]) : 0u;
  uint32_t total;
  uint32_t ex = block_excl_scan(cnt, s_tmp, &total);
  if (threadIdx.x < kChunks) s_cbase[threadIdx.x] = ex;
  __syncthreads();
  return total;
}

__device__ __forceinline__ uint32_t sample_rank(int j, const uint64_t *s_mask,
                                                const uint32_t *s_cbase) {
  int c = j * kWaves + (int)wave_id();
  return s_cbase[c] + (uint32_t)__popcll(s_mask[c] & lanemask_lt());
}

// True when the `count` composite keys (value<<16 | sample index) in s_keys[0..count) are
// not ascending.  Block-uniform result; contains barriers.
__device__ __forceinline__ bool keys_unsorted(const uint32_t *s_keys, uint32_t count,
                                              uint32_t *s_flag) {
  if (threadIdx.x == 0) *s_flag = 0u;
  __syncthreads();
  uint32_t bad = 0;
  for (uint32_t t = threadIdx.x; t + 1 < count; t += kBlock) bad |= (s_keys[t] > s_keys[t + 1]);
  if (bad) atomicOr(s_flag, 1u);
  __syncthreads();
  return *s_flag != 0u;
}

// Sort the keys (unique, so the order is the stable (value, input index) order), then turn
// the array into the inverse map s_keys[sample index] = sorted position.  `count` <= 32768.
// Only called for unsorted input: already ascending scans (the usual case) never get here.
//
// Angles of a scan are spread over the circle, so a counting sort on the top 11 bits of the
// angle word (2048 buckets, ~16 keys each for a full scan) followed by ranking every key inside
// its bucket by counting does the job in a few LDS passes; the keys wait in registers while
// `s_keys` is reused as the bucket-grouped array.  A scan that piles more than kBucketMax keys
// into one bucket (many equal angles) takes the bitonic network instead (120 stages).
constexpr uint32_t kSortBuckets = 2048;
constexpr uint32_t kBucketMax = 512;

struct SortLds {
  uint32_t start[kSortBuckets];
  uint32_t fill[kSortBuckets];
  uint32_t tmp[32];
};

__device__ __forceinline__ void sort_keys_to_positions(uint32_t *s_keys, uint32_t count,
                                                       SortLds &S) {
  uint32_t mine[kIters];
#pragma unroll
  for (int k = 0; k < kIters; ++k) {
    const uint32_t r = threadIdx.x + (uint32_t)k * kBlock;
    mine[k] = (r < count) ? s_keys[r] : 0xFFFFFFFFu;
  }
  for (uint32_t t = threadIdx.x; t < kSortBuckets; t += kBlock) S.start[t] = 0u;
  if (threadIdx.x == 0) S.tmp[31] = 0u;
  __syncthreads();
#pragma unroll
  for (int k = 0; k < kIters; ++k)
    if (mine[k] != 0xFFFFFFFFu) atomicAdd(&S.start[mine[k] >> 21], 1u);
  __syncthreads();
  {
    const uint32_t c0 = S.start[2 * threadIdx.x], c1 = S.start[2 * threadIdx.x + 1];
    if (max(c0, c1) > kBucketMax) S.tmp[31] = 1u;  // benign race: every writer stores 1
    uint32_t tot;
    const uint32_t ex = block_excl_scan(c0 + c1, S.tmp, &tot);
    S.start[2 * threadIdx.x] = ex;
    S.start[2 * threadIdx.x + 1] = ex + c0;
    S.fill[2 * threadIdx.x] = ex;
    S.fill[2 * threadIdx.x + 1] = ex + c0;
  }
  __syncthreads();
  if (S.tmp[31]) {  // block-uniform: degenerate distribution, s_keys is still untouched
    const uint32_t N = next_pow2(count);
    for (uint32_t t = count + threadIdx.x; t < N; t += kBlock) s_keys[t] = 0xFFFFFFFFu;
    block_bitonic_sort(s_keys, N);  // starts and ends with a barrier
#pragma unroll
    for (int k = 0; k < kIters; ++k) {
      const uint32_t r = threadIdx.x + (uint32_t)k * kBlock;
      mine[k] = (r < count) ? s_keys[r] : 0xFFFFFFFFu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kIters; ++k) {
      const uint32_t r = threadIdx.x + (uint32_t)k * kBlock;
      if (r < count) s_keys[mine[k] & 0xFFFFu] = r;
    }
    __syncthreads();
    return;
  }
  // group by bucket (order inside a bucket is arbitrary, the ranks below fix it)
#pragma unroll
  for (int k = 0; k < kIters; ++k)
    if (mine[k] != 0xFFFFFFFFu) s_keys[atomicAdd(&S.fill[mine[k] >> 21], 1u)] = mine[k];
  __syncthreads();
  // rank inside the bucket; the key is then replaced by (sample index << 16 | position), so no
  // second register array is needed
#pragma unroll
  for (int k = 0; k < kIters; ++k) {
    if (mine[k] != 0xFFFFFFFFu) {
      const uint32_t bkt = mine[k] >> 21;
      const uint32_t s0 = S.start[bkt], s1 = S.fill[bkt];
      uint32_t rank = s0;
      for (uint32_t q = s0; q < s1; ++q) rank += s_keys[q] < mine[k] ? 1u : 0u;
      mine[k] = (mine[k] << 16) | rank;  // rank < 32768: never collides with the empty marker
    }
  }
  __syncthreads();  // every bucket was read before the array is rewritten
#pragma unroll
  for (int k = 0; k < kIters; ++k)
    if (mine[k] != 0xFFFFFFFFu) s_keys[mine[k] >> 16] = mine[k] & 0xFFFFu;
  __syncthreads();
}

// ------------------------------------------------------------------------------
// k_ascend — SDK ascendScanData, in place (one workgroup per scan)
// ------------------------------------------------------------------------------
__device__ __forceinline__ float q14_to_deg(uint32_t q) {  // getAngle, :102-105 (exact in fp32)
  return (float)q * 90.f / 16384.f;
}
__device__ __forceinline__ uint32_t deg_to_q14(float v) {  // setAngle, :107-110 (u32 then u16 store)
  return ((uint32_t)(v * 16384.f / 90.f)) & 0xFFFFu;
}

// Two instantiations, two kernels: SORT = false handles the usual case (the scan comes out
// ascending once the invalid samples got their angles) and only flags a scan that needs
// sorting; SORT = true runs after it, returns at once for unflagged scans and does the whole
// job including the sort for the others.  Keeping the sort (32 keys per thread in registers)
// out of the first kernel keeps its register allocation spill-free.
// (Round 3: the SORT = false instance serves only calls of a few scans — k_ascend_stream below does
// the batches — and the SORT = true instance is a persistent grid over the LIST of flagged scans that
// k_ascend_stream appends to: need_sort[0] = how many, need_sort[1 ..] = which.  Launching one
// 1024-thread, 128 KiB workgroup per scan only to find its flag clear cost more than the
// streaming kernel itself.)
template <bool SORT>
__device__ __forceinline__ void ascend_one(uint32_t b, uint2 *__restrict__ nodes, uint32_t n_stride,
                                           const uint32_t *__restrict__ n_per_scan,
                                           uint32_t *__restrict__ status,
                                           uint32_t *__restrict__ need_sort, uint32_t *s_keys,
                                           uint32_t *s_misc, SortLds &s_sort, uint32_t mark = 0u,
                                           bool prefilled = false, uint32_t wrapped = 0u);

// mark: a status bit the queueing kernels set on a scan they hand to the sorting kernel — a
// single-scan call launches that kernel only when the bit came back (kAscendUnsorted, internal)
template <bool SORT>
__global__ __launch_bounds__(kBlock) void k_ascend(uint2 *__restrict__ nodes, uint32_t n_stride,
                                                   const uint32_t *__restrict__ n_per_scan,
                                                   uint32_t *__restrict__ status,
                                                   uint32_t *__restrict__ need_sort, uint32_t mark,
                                                   uint32_t prefilled, uint32_t *__restrict__ sort_stat) {
  __shared__ uint32_t s_keys[kMaxN];
  __shared__ uint32_t s_misc[8];
  __shared__ SortLds s_sort;
  if (SORT) {
    const uint32_t count = need_sort[0];
    // (developer aid: how many scans of this call took the sort — rplgpu_debug_ascend_sorted)
    if (sort_stat && blockIdx.x == 0 && threadIdx.x == 0) *sort_stat = count;
    // A grid of ONE workgroup (the single-scan call) leaves the list empty for the next call
    // itself, so that such a call needs no clearing command in front of it.
    if (gridDim.x == 1u) {
      __syncthreads();  // (every thread has read the count)
      if (threadIdx.x == 0) need_sort[0] = 0u;
    }
    for (uint32_t k = blockIdx.x; k < count; k += gridDim.x) {
      // (a list entry: scan index | wrapped fills at the scan's front << 24, see k_ascend_stream)
      const uint32_t entry = need_sort[1u + k];
      ascend_one<SORT>(entry & 0x00FFFFFFu, nodes, n_stride, n_per_scan, status, need_sort, s_keys,
                       s_misc, s_sort, 0u, prefilled != 0u, entry >> 24);
      __syncthreads();  // LDS is reused by the next scan
    }
  } else {
    ascend_one<SORT>(blockIdx.x, nodes, n_stride, n_per_scan, status, need_sort, s_keys, s_misc, s_sort, mark);
  }
}

template <bool SORT>
__device__ __forceinline__ void ascend_one(uint32_t b, uint2 *__restrict__ nodes, uint32_t n_stride,
                                           const uint32_t *__restrict__ n_per_scan,
                                           uint32_t *__restrict__ status,
                                           uint32_t *__restrict__ need_sort, uint32_t *s_keys,
                                           uint32_t *s_misc, SortLds &s_sort, uint32_t mark,
                                           bool prefilled, uint32_t wrapped) {
  const uint32_t n = min(n_per_scan[b], min(n_stride, kMaxN));  // never past the slot
  uint2 *scan = nodes + (size_t)b * n_stride;

  uint2 v[kIters];
  load_scan(scan, n, v);
  if (SORT && prefilled) {
    // The scan comes from k_ascend_stream: every filled angle word is already in place (that kernel
    // writes the fills of a scan it queues, too) and nodes may have MOVED since (its local repair),
    // so an invalid node's index no longer says which angle it was given — the stored words are the
    // truth.  What is left of :171-181 is the sort; ties stay in the order they are in (the repair
    // is a stable permutation, so that is still the input order) — except for the `wrapped` fills
    // that kernel put at the scan's FRONT (:174-176: an angle near zero): they are the LAST nodes of
    // the input, every other node sits `wrapped` places up, and a fill must rank behind a node of
    // equal angle.  The key's tie field is therefore the place in the INPUT, not in the buffer.
    wrapped = min(wrapped, n);
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t i = sample_index(j);
      if (i < n) s_keys[i] = (nd_q14(v[j]) << 16) | (i < wrapped ? n - wrapped + i : i - wrapped);
    }
    // (sort_keys_to_positions leaves the final position of the node whose tie field is t in s_keys[t])
    __syncthreads();
    sort_keys_to_positions(s_keys, n, s_sort);
    load_scan(scan, n, v);
    // every thread has its samples back IN REGISTERS before anybody overwrites the buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t i = sample_index(j);
      if (i < n) {
        const uint32_t pos = s_keys[i < wrapped ? n - wrapped + i : i - wrapped];
        if (pos != i) scan[pos] = v[j];
      }
    }
    return;
  }

  // first valid index (block min)
  if (threadIdx.x == 0) s_misc[0] = 0xFFFFFFFFu;
  __syncthreads();
  uint32_t first = 0xFFFFFFFFu;
#pragma unroll
  for (int j = kIters - 1; j >= 0; --j) {
    uint32_t i = sample_index(j);
    if (i < n && nd_dist(v[j]) != 0u) first = i;
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) first = min(first, (uint32_t)__shfl_xor((int)first, d, 64));
  if (lane_id() == 0 && first != 0xFFFFFFFFu) atomicMin(&s_misc[0], first);
  __syncthreads();
  first = s_misc[0];
  if (first == 0xFFFFFFFFu) {  // :151 all invalid -> SL_RESULT_OPERATION_FAIL, buffer untouched
    if (threadIdx.x == 0 && status) status[b] = RPLGPU_SCAN_ALL_INVALID;
    return;  // (not queued for the sorting kernel)
  }
  if (threadIdx.x == 0 && status) status[b] = 0u;

  const float inc = 360.f / (float)n;  // :131

  // head chain (:135-148): a quantised recurrence from the first valid sample back to
  // sample 0. The fill pass below overwrites every invalid sample i>=1, and the tail
  // pass (:154-168) only touches such samples, so only sample 0's value survives.
  if (threadIdx.x == 0) {
    uint32_t q = nd_q14(scan[first]);
    for (uint32_t s = first; s > 0; --s) {
      float e = q14_to_deg(q) - inc;
      if (e < 0.0f) e = 0.0f;
      q = deg_to_q14(e);
    }
    s_misc[1] = q;  // angle of sample 0 after head tuning (unchanged when first == 0)
  }
  __syncthreads();
  const uint32_t front_q = s_misc[1];
  const float front = q14_to_deg(front_q);  // :171

  // the fill pass (:171-178) for the 32 samples of this thread: new angle words in v, keys out
  uint32_t changed = 0;
  auto fill_angles = [&](bool write_keys) {
    changed = 0;
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      uint32_t i = sample_index(j);
      if (i < n) {
        uint32_t q = nd_q14(v[j]);
        uint32_t nq = q;
        if (nd_dist(v[j]) == 0u) {
          if (i == 0) {
            nq = front_q;
          } else {  // :172-178
            float e = front + (float)i * inc;
            if (e > 360.0f) e -= 360.0f;
            nq = deg_to_q14(e);
          }
        }
        if (nq != q) {
          changed |= 1u << j;
          v[j].x = (v[j].x & 0xFFFF0000u) | nq;
        }
        if (write_keys) s_keys[i] = (nq << 16) | i;  // unique key: (angle, input index)
      }
    }
  };
  fill_angles(true);
  __syncthreads();

  // std::sort by float angle (:181) == sort by q14 (getAngle is exact and monotone).
  // Equal angles: the reference order is whatever introsort leaves; ours is input order.
  if (!SORT) {
    const bool unsorted = keys_unsorted(s_keys, n, &s_misc[2]);
    // not ascending: queue the scan for the sorting kernel (need_sort[0] = count, then the list)
    if (threadIdx.x == 0 && unsorted) {
      need_sort[1u + atomicAdd(&need_sort[0], 1u)] = b;
      if (mark && status) status[b] |= mark;
    }
    if (!unsorted) {  // already ascending: only the filled angles move
#pragma unroll
      for (int j = 0; j < kIters; ++j)
        if ((changed >> j) & 1u) scan[sample_index(j)] = v[j];
    }
    return;  // unsorted: nothing written, the second kernel redoes this scan
  }
  // the sort keeps 32 keys per thread in registers: the scan is dropped and fetched again (L2)
  // rather than spilled around it — this path is the rare one.  Every thread has its samples
  // back in registers before anybody overwrites the buffer.
  sort_keys_to_positions(s_keys, n, s_sort);
  load_scan(scan, n, v);
  fill_angles(false);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    uint32_t i = sample_index(j);
    if (i < n) {
      uint32_t pos = s_keys[i];
      if (pos != i || ((changed >> j) & 1u)) scan[pos] = v[j];
    }
  }
}

// ------------------------------------------------------------------------------
// k_ascend_stream — the usual case of ascendScanData as a STREAMING kernel (round 3).
// k_ascend<false> above keeps a whole scan in registers and its keys in 128 KiB of LDS: one
// workgroup per compute unit, and the load of a scan, the work on it and its stores never
// overlap (0.32 ms per 4096 x 32 000 samples, 41 % of the roofline).  What the usual case needs is
// far less: the first valid sample, the head chain (:135-148) when that is not sample 0, and
// then one elementwise pass — the interpolated angle of an invalid sample i depends on (front, i,
// inc) only (:171-178) — that also checks that the angle words come out non-descending.  So:
// 256-thread workgroups (eight per compute unit), 16-byte loads four deep, 8-byte stores of the
// changed samples only, neighbours compared in registers / by DPP / through 2 words of LDS per
// 128-sample chunk.  A scan that is not ascending is flagged for k_ascend<true> exactly as before;
// that kernel recomputes every filled angle from the same (front, i, inc), so the fills this
// kernel already wrote in place do not disturb it.
// ------------------------------------------------------------------------------
constexpr int kAscT = 256;
constexpr int kAscW = kAscT / 64;
typedef uint32_t asc_u32x4 __attribute__((ext_vector_type(4)));

// Local repair (round 5).  On a real sensor the interpolated angles of the invalid nodes do not mesh
// with their measured neighbours, and measured angles jitter: the filled scan is ALMOST ascending —
// neighbouring samples swapped, every node within a few places of where :181's sort puts it — and
// sending every such scan through k_ascend<true> (a full counting sort of 32 000 keys by one
// workgroup) cost 8 x the streaming pass.  The streaming kernel now repairs local disorder itself:
//   phase A  a 128-sample chunk (one wave trip) whose angle words are not ascending goes through a
//            few rounds of odd-even transposition on 32-bit tokens (angle word << 8 | slot in the
//            chunk) held two per lane — min / max inside the lane, then against the neighbour
//            lanes by DPP — and the nodes follow their tokens through 1 KiB of wave-private LDS;
//   phase B  after the scan's last chunk: a chunk boundary whose two sides are out of order
//            (last word of chunk k - 1 > first word of chunk k) is repaired in the 32-sample window
//            across it, one sample per lane, two windows per wave, odd-even transposition by DPP
//            wave shifts, the nodes following by ds_bpermute.
// Both are stable permutations (a swap only where the left token is strictly larger, tokens carry
// the original place), nothing is ever dropped, and the result is CHECKED: every chunk must come out
// ascending, every repaired window too, with its two end samples still in place (they anchor the
// window to the sorted chunks around it).  A scan that fails a check — disorder reaching farther than
// the rounds — is queued for k_ascend<true> as before, which then trusts the stored angle words
// (`prefilled`).  Ties keep their input order, the library's tie rule (tests/canon.py).
constexpr int kAscRounds = 128;  // phase A: at most this many (inside the lane, across lanes) rounds: sorts any chunk
constexpr int kAscRoundsB = 16;  // phase B: at most this many pairs of (even, odd) rounds on 32 samples

template <int CTRL>
__device__ __forceinline__ uint32_t asc_dpp_mov(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xF, 0xF, false);
}

// (eight waves per SIMD = 64 registers: the streaming pass is paced by the loads it keeps in flight;
// round 6's merges would otherwise take 82 registers and the occupancy down to five)
__global__ __launch_bounds__(kAscT) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_ascend_stream(uint2 *__restrict__ nodes, uint32_t n_stride,
                                                         const uint32_t *__restrict__ n_per_scan,
                                                         uint32_t *__restrict__ status,
                                                         uint32_t *__restrict__ need_sort, uint32_t mark) {
  __shared__ uint32_t s_misc[8];                  // 0 first valid, 1 front word, 2 not ascending, 3 wrap zone, 4 W, 5-6 its invalid nodes
  __shared__ uint32_t s_edge[2 * (kMaxN / 128)];  // first / last angle word of every 128-sample chunk
  __shared__ uint4 s_xch[kAscW][64];              // phase A: the nodes of a chunk, by slot (wave-private)
  __shared__ uint2 s_wrap[64];                    // the wrapped fills at the scan's end (they go to its front)
  __shared__ uint32_t s_tok[kAscW][256];          // phase C: the tokens of a pair of chunks (wave-private)
  const uint32_t b = blockIdx.x;
  const uint32_t n = (uint32_t)__builtin_amdgcn_readfirstlane(
      (int)min(n_per_scan[b], min(n_stride, kMaxN)));  // never past the slot
  uint2 *scan = nodes + (size_t)b * n_stride;
  // bounds-checked 16-byte loads (dword alignment suffices; beyond the scan: zeros = invalid)
  const __amdgpu_buffer_rsrc_t rsrc =
      __builtin_amdgcn_make_buffer_rsrc((void *)scan, 0, (int)(n * 8u), 0x00020000);
  auto load_pair = [&](uint32_t pair) -> uint4 {
    const asc_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(pair * 16u), 0, 0);
    return make_uint4(t.x, t.y, t.z, t.w);
  };
  const uint32_t npairs = (n + 1u) >> 1;
  if (threadIdx.x == 0) { s_misc[0] = 0xFFFFFFFFu; s_misc[2] = 0u; s_misc[4] = 0u; }
  __syncthreads();
  // ---- first valid sample (block-uniform loop; found in the first round for any real scan)
  for (uint32_t base = 0; base < npairs; base += kAscT) {
    const uint4 w = load_pair(base + threadIdx.x);
    const uint32_t i = 2u * (base + threadIdx.x);
    uint32_t cand = 0xFFFFFFFFu;
    if (i + 1u < n && nd_dist(make_uint2(w.z, w.w)) != 0u) cand = i + 1u;
    if (i < n && nd_dist(make_uint2(w.x, w.y)) != 0u) cand = i;
    const uint64_t any = __builtin_amdgcn_ballot_w64(cand != 0xFFFFFFFFu);
    if (any && lane_id() == (uint32_t)__builtin_ctzll(any)) atomicMin(&s_misc[0], cand);
    __syncthreads();
    // every wave reads the round's result BEFORE any wave may write the next round's (a wave that
    // saw "none yet" and ran ahead into the next atomicMin could otherwise make a slower wave
    // leave the loop one round early: mismatched barriers, front_q read before it is written)
    const uint32_t found = s_misc[0];
    __syncthreads();
    if (found != 0xFFFFFFFFu) break;
  }
  const uint32_t first = s_misc[0];
  if (first == 0xFFFFFFFFu) {  // :151 all invalid -> SL_RESULT_OPERATION_FAIL, buffer untouched
    if (threadIdx.x == 0 && status) status[b] = RPLGPU_SCAN_ALL_INVALID;
    return;
  }
  const float inc = 360.f / (float)n;  // :131
  if (threadIdx.x == 0) {
    if (status) status[b] = 0u;
    // head chain (:135-148), see k_ascend: only sample 0's value survives the fill pass
    uint32_t q = nd_q14(scan[first]);
    for (uint32_t s = first; s > 0; --s) {
      float e = q14_to_deg(q) - inc;
      if (e < 0.0f) e = 0.0f;
      q = deg_to_q14(e);
    }
    s_misc[1] = q;
    // The wrap zone: the last indices whose interpolated angle passes 360 degrees (:174-176).  An
    // invalid node there is given an angle near ZERO and belongs at the FRONT of the sorted scan —
    // every other node then moves up by one place, which no local repair can do.  (front + i * inc
    // is non-decreasing in i: the zone is a suffix of the index range; 65 = longer than handled.)
    const float fr = q14_to_deg(q);
    uint32_t tl = 0u;
    while (tl < 65u && tl + 1u < n) {
      const float e = fr + (float)(n - 1u - tl) * inc;
      if (!(e > 360.0f)) break;
      ++tl;
    }
    s_misc[3] = tl;
  }
  __syncthreads();
  const uint32_t front_q = s_misc[1];
  const float front = q14_to_deg(front_q);  // :171
  // W: the wrap zone — the last tl indices — holds W invalid nodes, the wrapped fills.  This kernel moves
  // them itself (to the front, in index order: their angles ascend; everything else up by W) when the
  // zone lies inside the scan's LAST 128-sample chunk: there the fills count as "larger than any angle
  // word", phase A lets them sink behind the chunk's other nodes, and what is stored is what is left.
  // A longer zone, or one that reaches into the chunk before, is left to the checks below (the scan
  // then comes out not ascending and goes to the sorting kernel).
  {
    const uint32_t tl = s_misc[3];
    if (tl >= 1u && tl <= 64u && n - tl >= ((n - 1u) & ~127u) && threadIdx.x < 64u) {  // wave 0
      const bool inv = lane_id() < tl && nd_dist(scan[n - tl + lane_id()]) == 0u;
      const uint64_t m = __builtin_amdgcn_ballot_w64(inv);  // bit j: index n - tl + j
      if (lane_id() == 0u) {
        s_misc[4] = (uint32_t)__popcll(m);
        s_misc[5] = (uint32_t)m;
        s_misc[6] = (uint32_t)(m >> 32);
      }
    }
  }
  __syncthreads();
  const uint32_t W = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_misc[4]);
  const uint32_t zone0 = n - s_misc[3];  // first index of the wrap zone (only meaningful when W > 0)
  const uint64_t zmask = ((uint64_t)s_misc[6] << 32) | s_misc[5];
  const uint32_t nm = n - W;  // the other nodes keep their order relative to each other (up to local repair)
  auto wrapped = [&](uint32_t i) -> bool {  // an invalid node of the wrap zone
    return W && i >= zone0 && i < n && ((zmask >> (i - zone0)) & 1ull);
  };
  // ---- the fill pass (:171-178) + the order of the result (:181 moves nothing when it ascends)
  auto new_angle = [&](uint2 v, uint32_t i) -> uint32_t {
    uint32_t nq = nd_q14(v);
    if (nd_dist(v) == 0u) {
      if (i == 0u) {
        nq = front_q;
      } else {  // :172-178
        float e = front + (float)i * inc;
        if (e > 360.0f) e -= 360.0f;
        nq = deg_to_q14(e);
      }
    }
    return nq;
  };
  uint32_t bad = 0u;  // this lane saw an order violation that the repairs did not remove
  const uint32_t wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)wave_id());
  constexpr int kDeep = 4;  // loads in flight per thread
  constexpr uint32_t kTrip = (uint32_t)kDeep * kAscT;  // pairs per trip of the workgroup
  // With W > 0 the chunks are taken from the END of the scan to its front: a node is written W
  // places above where it was read, i.e. into places that were read in this trip (after the barrier
  // below) or in an earlier one.  (Front to end otherwise.)
  const uint32_t ntrips = (npairs + kTrip - 1u) / kTrip;
  for (uint32_t t = 0; t < ntrips; ++t) {
    const uint32_t base = (W ? ntrips - 1u - t : t) * kTrip;
    uint4 w[kDeep];
#pragma unroll
    for (int k = 0; k < kDeep; ++k) w[k] = load_pair(base + (uint32_t)k * kAscT + threadIdx.x);
    if (W) {  // block-uniform: every wave has its nodes in registers before any wave stores
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < kDeep; ++k) {
      const uint32_t pair = base + (uint32_t)k * kAscT + threadIdx.x, i = 2u * pair;
      const uint2 a = make_uint2(w[k].x, w[k].y), c = make_uint2(w[k].z, w[k].w);
      // the nodes with their new angle words
      const uint32_t fa = i < n ? new_angle(a, i) : 0u, fc = i + 1u < n ? new_angle(c, i + 1u) : 0u;
      uint2 na = make_uint2((a.x & 0xFFFF0000u) | fa, a.y);
      uint2 nc = make_uint2((c.x & 0xFFFF0000u) | fc, c.y);
      const bool wa = wrapped(i), wc = wrapped(i + 1u);
      if (W) {  // the wrapped fills wait in LDS (in index order) until the front of the scan has been read
        if (wa) s_wrap[__popcll(zmask & ((1ull << (i - zone0)) - 1ull))] = na;
        if (wc) s_wrap[__popcll(zmask & ((1ull << (i + 1u - zone0)) - 1ull))] = nc;
      }
      // (the wrapped fills and the samples beyond the scan compare as "larger than any angle word":
      // they end up behind the last chunk's other nodes and are not stored from there)
      uint32_t qa = (i < n && !wa) ? fa : 0x10000u;
      uint32_t qc = (i + 1u < n && !wc) ? fc : 0x10000u;
      // ---- phase A: is the chunk in order?  (lane l + 1's first word against this lane's second)
      const uint32_t nxt_q = asc_dpp_mov<0x130>(0xFFFFFFFFu, qa);  // wave_shl:1; lane 63: no successor here
      if (__builtin_amdgcn_ballot_w64((qa > qc) | (qc > nxt_q))) {  // wave-uniform: local repair
        uint32_t ta = (qa << 8) | (2u * lane_id()), tc = (qc << 8) | (2u * lane_id() + 1u);
        // rounds in pairs until the chunk is in order (wave-uniform exit: +-3 words of jitter need
        // two or three pairs; 128 rounds sort any 128 tokens — round 6: every chunk leaves this phase
        // SORTED, which is what the merges of phase C build on; and a wrapped fill left in front of a
        // node that stays would be stored in its place).  (Also built: a 28-stage bitonic network for
        // chunks the first 16 rounds do not finish — four inlined copies of it cost the uniform case
        // 14 % and +-64 words 14 %: profiles/r06/ascend_merge_r06.txt.)
        for (int r = 0; r < kAscRounds; r += 2) {
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const uint32_t lo = min(ta, tc), hi = max(ta, tc);
            const uint32_t nx = asc_dpp_mov<0x130>(0xFFFFFFFFu, lo);  // lane l + 1's smaller token
            const uint32_t pv = asc_dpp_mov<0x138>(0u, hi);           // lane l - 1's larger token (wave_shr:1)
            ta = max(lo, pv);
            tc = min(hi, nx);
          }
          const uint32_t nxa = asc_dpp_mov<0x130>(0xFFFFFFFFu, ta);
          if (!__builtin_amdgcn_ballot_w64((ta > tc) | (tc > nxa))) break;
        }
        {  // (the last exchange across lanes may leave a lane's two tokens swapped)
          const uint32_t lo = min(ta, tc), hi = max(ta, tc);
          ta = lo;
          tc = hi;
        }
        // the nodes follow their tokens (slot = low 7 bits) through the wave's 1 KiB of LDS
        s_xch[wv][lane_id()] = make_uint4(na.x, na.y, nc.x, nc.y);
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint2 *slots = reinterpret_cast<const uint2 *>(&s_xch[wv][0]);
        na = slots[ta & 127u];
        nc = slots[tc & 127u];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read before the next chunk overwrites it)
        qa = ta >> 8;
        qc = tc >> 8;
      }
      {
        const bool sa = i < nm && (W || na.x != a.x || na.y != a.y);
        const bool sc = i + 1u < nm && (W || nc.x != c.x || nc.y != c.y);
        // (a run of invalid nodes changes both nodes of most of its pairs: one 16-byte store then — the
        // pair is 16-byte aligned when W is even)
        // Whole 128-byte lines (eight lanes' pairs) wherever one of their nodes changed (round 6: partial-line
        // writes of scattered 8- and 16-byte pieces cost +-3 words of jitter 9 %); a wave with nothing to
        // store skips all of it.
        const uint64_t chg = __builtin_amdgcn_ballot_w64(sa || sc);
        if (chg != 0ull) {  // wave-uniform
          uint64_t g = W ? 0ull : chg;
          g |= g >> 4;
          g |= g >> 2;
          g |= g >> 1;
          g = (g & 0x0101010101010101ull) * 0xFFull;
          if (__builtin_amdgcn_inverse_ballot_w64(g) && i + 1u < nm) {
            *reinterpret_cast<uint4 *>(&scan[i]) = make_uint4(na.x, na.y, nc.x, nc.y);
          } else if (sa && sc && !(W & 1u)) {
            *reinterpret_cast<uint4 *>(&scan[i + W]) = make_uint4(na.x, na.y, nc.x, nc.y);
          } else {
            if (sa) scan[i + W] = na;
            if (sc) scan[i + 1u + W] = nc;
          }
        }
      }
      // what is left out of order inside the chunk after the repair (nothing, normally)
      const uint32_t prev = asc_dpp_mov<0x138>(0u, qc);  // the lane to the left (lane 0: nothing before it here)
      bad |= (qa > qc) | (prev > qa);
      const uint32_t chunk = pair >> 6;  // wave-uniform: 64 pairs = 128 samples
      if (pair < npairs) {
        if (lane_id() == 0u) s_edge[2u * chunk] = qa;
        if (lane_id() == 63u) s_edge[2u * chunk + 1u] = qc;
      }
    }
  }
  __syncthreads();  // (also: the nodes every wave wrote are visible to the whole workgroup)
  // ---- phase B: chunk boundaries whose two sides are out of order
  const uint32_t nchunks = (npairs + 63u) >> 6;
  const uint32_t half = lane_id() >> 5, col = lane_id() & 31u;  // two 32-sample windows per wave
  uint32_t win_failed = 0u;  // a window of this lane's could not repair its boundary
  for (uint32_t base = 1u; base < nchunks; base += (uint32_t)kAscT / 32u) {
    const uint32_t kb = base + wv * 2u + half;  // the boundary between chunk kb - 1 and chunk kb
    const bool need = kb < nchunks && s_edge[2u * kb - 1u] > s_edge[2u * kb];
    if (!__builtin_amdgcn_ballot_w64(need)) continue;  // wave-uniform
    const uint32_t pos = 128u * kb - 16u + col;  // (place among the nm nodes; W above that in the buffer)
    uint2 v = make_uint2(0u, 0u);
    if (need && pos < nm) v = scan[pos + W];
    const uint32_t q = (need && pos < nm) ? nd_q14(v) : 0x10000u;
    uint32_t t = (q << 8) | col;
    const bool odd = (col & 1u) != 0u;
    for (int r = 0; r < kAscRoundsB; ++r) {
      {  // pairs (0,1) (2,3) ...: the even lane keeps the smaller token
        const uint32_t nx = asc_dpp_mov<0x130>(t, t);  // wave_shl:1 (an odd lane never uses it)
        const uint32_t pv = asc_dpp_mov<0x138>(t, t);  // wave_shr:1 (an even lane never uses it)
        t = odd ? max(t, pv) : min(t, nx);
      }
      {  // pairs (1,2) (3,4) ...: the odd lane keeps the smaller token; a window's end lanes stay
        const uint32_t nx = asc_dpp_mov<0x130>(t, t);
        const uint32_t pv = asc_dpp_mov<0x138>(t, t);
        const uint32_t t2 = odd ? min(t, nx) : max(t, pv);
        t = (col == 0u || col == 31u) ? t : t2;
      }
      const uint32_t pv = asc_dpp_mov<0x138>(0u, t);
      if (!__builtin_amdgcn_ballot_w64(col != 0u && pv > t)) break;  // wave-uniform: both windows ascend
    }
    // checks: the window ascends, its end samples did not move.  A window that fails (disorder that
    // reaches farther than 16 places) is NOT stored: both chunks stay sorted, their edges stay out of
    // order, and phase C merges the two chunks.
    const uint32_t pv = asc_dpp_mov<0x138>(0u, t);
    const bool viol = need & ((col != 0u && pv > t) | ((col == 0u || col == 31u) && (t & 31u) != col));
    const uint64_t vm = __builtin_amdgcn_ballot_w64(viol);
    const bool win_ok = ((vm >> (lane_id() & 32u)) & 0xFFFFFFFFull) == 0ull;  // (this half-wave's window)
    // the nodes follow: source lane = same window, column = the token's low bits
    const uint32_t src = ((lane_id() & 32u) | (t & 31u)) * 4u;
    const uint32_t mx = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)v.x);
    const uint32_t my = (uint32_t)__builtin_amdgcn_ds_bpermute((int)src, (int)v.y);
    if (need && win_ok) {
      if (pos < nm && (t & 31u) != col) scan[pos + W] = make_uint2(mx, my);
      if (col == 15u) s_edge[2u * kb - 1u] = t >> 8;  // the chunks' new edge words
      if (col == 16u) s_edge[2u * kb] = t >> 8;
    }
    win_failed |= need && !win_ok;
  }
  // ---- phase C (round 6): chunk boundaries that are still out of order — disorder beyond the
  // windows' 16 places — are repaired by MERGING the two sorted chunks: odd boundaries first (pairs
  // (0,1) (2,3) ...), then even ones ((1,2) (3,4) ...).  Two such passes sort anything whose
  // nodes are at most 128 places from where they belong (every chunk is sorted when it gets here);
  // only what is still out of order afterwards goes to the one-workgroup sorting kernel.  A wave
  // merges two sorted runs A, B of up to 128 nodes that follow each other in the buffer: 256 tokens
  // (angle word << 8 | tie field: unique) in LDS, every node's new place = its place in its own run +
  // the number of smaller tokens in the other (a binary search), nodes stored where they moved.
  // Tie field: A's nodes before B's (the input order of two chunks) — or behind them, for the
  // wrapped fills (run A, at the scan's front), which are the LAST nodes of the input.
  // (the barrier that publishes phase B's stores and edge words also tells every wave whether any
  // boundary is left: a boundary phase B did not look at was in order, one it repaired is now)
  const bool boundaries_left = __syncthreads_or((int)win_failed) != 0;
  auto merge_runs = [&](uint32_t a0, uint32_t la, uint32_t lb, bool a_last, uint32_t edge0) {
    // wave-uniform arguments; a0: A's first place in the buffer; edge0: s_edge index of A's first word
    // (chunk merges only: la == 128) or 0xFFFFFFFF
    const uint32_t e0 = 4u * lane_id();  // tokens e0 .. e0 + 3: run A in the slots 0-127, run B in 128-255
    uint2 v[4];
    uint32_t t[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t e = e0 + (uint32_t)r, k = e & 127u;
      const bool in_a = e < 128u, live = in_a ? k < la : k < lb;
      v[r] = live ? scan[a0 + (in_a ? k : la + k)] : make_uint2(0u, 0u);
      const uint32_t tie = (in_a != a_last) ? k : 128u + k;
      t[r] = ((live ? nd_q14(v[r]) : 0x10000u) << 8) | tie;  // (an empty slot: behind everything)
    }
    uint32_t *tok = s_tok[wv];
    *reinterpret_cast<uint4 *>(&tok[e0]) = make_uint4(t[0], t[1], t[2], t[3]);
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const uint32_t other = lane_id() < 32u ? 128u : 0u;  // the other run's tokens
    uint32_t place[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      uint32_t c = 0u;  // tokens of the other run below mine
#pragma unroll
      for (uint32_t st = 64u; st >= 1u; st >>= 1) c += tok[other + c + st - 1u] < t[r] ? st : 0u;
      c += tok[other + c] < t[r] ? 1u : 0u;  // (c <= 127 here)
      place[r] = ((e0 + (uint32_t)r) & 127u) + c;  // place in the merged run
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (read before the next merge's tokens overwrite them)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const uint32_t e = e0 + (uint32_t)r, k = e & 127u;
      const bool in_a = e < 128u, live = in_a ? k < la : k < lb;
      if (live) {
        if (place[r] != (in_a ? k : la + k)) scan[a0 + place[r]] = v[r];
        if (edge0 != 0xFFFFFFFFu) {  // the two chunks' new edge words
          const uint32_t word = t[r] >> 8;
          if (place[r] == 0u) s_edge[edge0] = word;
          if (place[r] == 127u) s_edge[edge0 + 1u] = word;
          if (place[r] == 128u) s_edge[edge0 + 2u] = word;
          if (place[r] == 255u) s_edge[edge0 + 3u] = word;
        }
      }
    }
  };
  auto boundaries_bad = [&]() -> bool {  // block-uniform: some chunk boundary is out of order (one barrier)
    uint32_t any = 0u;
    for (uint32_t kb = 1u + threadIdx.x; kb < nchunks; kb += (uint32_t)kAscT)
      any |= s_edge[2u * kb - 1u] > s_edge[2u * kb];
    return __syncthreads_or((int)any) != 0;
  };
  if (boundaries_left) {
    for (uint32_t first_kb = 1u; first_kb <= 2u; ++first_kb) {  // odd boundaries, then even ones
      for (uint32_t kb = first_kb + 2u * wv; kb < nchunks; kb += 2u * (uint32_t)kAscW)
        if (s_edge[2u * kb - 1u] > s_edge[2u * kb])  // wave-uniform
          merge_runs(W + 128u * (kb - 1u), 128u, min(128u, nm - 128u * kb), false, 2u * kb - 2u);
      __syncthreads();  // (the merged nodes and edge words, before the other parity reads them)
    }
    if (boundaries_bad()) bad |= 1u;
  }
  // ---- the wrapped fills: to the front, in index order (their angles ascend).  Nothing is stored
  // before the workgroup's verdict: a scan that goes to the sorting kernel after all must reach it as
  // "W fills in index order, then every other node W places up" — that kernel's tie rule counts on it
  // (ADVICE r5).  A fill belongs behind every node of smaller OR EQUAL word (ties keep the input
  // order and the fills are the input's last nodes): when the nodes at the scan's front reach that
  // far — jittered angles — the fills and the first chunk are one more merge.  It must not carry a
  // fill past the first chunk: then the scan is the sorting kernel's.
  if (W && nchunks > 1u && threadIdx.x == 0u && nd_q14(s_wrap[W - 1u]) >= s_edge[2]) bad |= 1u;
  if (__builtin_amdgcn_ballot_w64(bad != 0u) && lane_id() == 0u) atomicOr(&s_misc[2], 1u);
  __syncthreads();
  const bool unsorted = s_misc[2] != 0u;  // block-uniform
  if (W) {
    if (threadIdx.x < W) scan[threadIdx.x] = s_wrap[threadIdx.x];  // (the nodes behind them are W places up already)
    if (!unsorted && nd_q14(s_wrap[W - 1u]) >= s_edge[0]) {  // block-uniform
      __syncthreads();  // (the fills are in the buffer)
      if (wv == 0u) merge_runs(0u, W, min(128u, nm), true, 0xFFFFFFFFu);
    }
  }
  // still not ascending: queue the scan for the sorting kernel (need_sort[0] = count, then the list:
  // scan index | wrapped fills at its front << 24)
  if (threadIdx.x == 0 && unsorted) {
    need_sort[1u + atomicAdd(&need_sort[0], 1u)] = b | (W << 24);
    if (mark && status) status[b] |= mark;
  }
}

// ------------------------------------------------------------------------------
// k_laserscan_raw — publish_scan Mode B, scan_processing = false (one workgroup per scan);
// Mode A lives in rpl_laserscan.hip
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_laserscan_raw(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, float *__restrict__ ranges, float *__restrict__ intens,
    uint32_t *__restrict__ beam_count) {
  __shared__ uint32_t s_keys[kMaxN];
  __shared__ uint64_t s_mask[kChunks];
  __shared__ uint32_t s_cbase[kChunks];
  __shared__ uint32_t s_tmp[32];
  __shared__ SortLds s_sort;

  const uint32_t b = blockIdx.x;
  const uint32_t n = min(n_per_scan[b], min(n_stride, kMaxN));  // never past the slot
  const uint2 *scan = nodes + (size_t)b * n_stride;
  float *out_r = ranges + (size_t)b * n_stride;
  float *out_i = intens + (size_t)b * n_stride;

  uint2 v[kIters];
  load_scan(scan, n, v);

  // LOOP 1 (:583-602): keep mask; the unit conversions are redone where needed.
  uint32_t kept = 0;
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    uint32_t d = nd_dist(v[j]);
    if (nd_keep(d, nd_quality(v[j]), p)) kept |= 1u << j;
  }
  const uint32_t count = publish_masks_and_scan(kept, s_mask, s_cbase, s_tmp);
  if (threadIdx.x == 0) beam_count[b] = count;
  if (count == 0) return;  // :611-613 nothing published

  // ---- Mode B (:663-680): sorted order, reversed unless inverted.
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    if ((kept >> j) & 1u) {
      s_keys[sample_rank(j, s_mask, s_cbase)] = (nd_q14(v[j]) << 16) | sample_index(j);
    }
  }
  __syncthreads();
  const bool unsorted = keys_unsorted(s_keys, count, &s_tmp[20]);
  if (unsorted) {
    // the sort keeps 32 keys per thread in registers: the scan is dropped and fetched again
    // (L2) rather than spilled around it — this path is the rare one
    sort_keys_to_positions(s_keys, count, s_sort);
    load_scan(scan, n, v);
  }
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    if ((kept >> j) & 1u) {
      // ascending input: the sorted position is the rank among the kept samples
      uint32_t pos = unsorted ? s_keys[sample_index(j)] : sample_rank(j, s_mask, s_cbase);
      uint32_t idx = p.inverted ? pos : (count - 1u - pos);  // :676
      out_r[idx] = nd_dist_m(nd_dist(v[j]));
      out_i[idx] = nd_intensity(nd_quality(v[j]), p.is_new_protocol);
    }
  }
}

// ------------------------------------------------------------------------------
// k_cloud — E1 keep mask + E2 polar->XYZ, stable compaction in input order
// (Round 5 measured it on the C3 batch for the first time: 0.63-0.645 ms per 4096 x 32 000 samples,
// 118.6 M points out = 8 B read + 14.5 B written per sample at 4.6 TB/s, 0.57-0.585 of the 8 TB/s
// roofline, i.e. 73 % of the chip's copy rate: write bound.  A streaming form — 256-thread
// workgroups, a running count, chunk counts through LDS, points staged through a wave-private LDS
// tile so that every store instruction writes 64 consecutive points — was built, parity green, and
// measured 0.652-0.668 ms: no gain over holding the scan in registers, removed.  Its first version
// let lane l store its own two samples, i.e. every other 16-byte point per instruction: 1.36 ms.)
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_cloud(
    const uint2 *__restrict__ nodes, uint32_t n_stride, const uint32_t *__restrict__ n_per_scan,
    KParams p, Tables T, const uint32_t *__restrict__ keepmask, uint32_t mask_stride,
    float4 *__restrict__ xyzi, uint32_t out_stride, uint32_t *__restrict__ n_points,
    uint32_t *__restrict__ status, const float *__restrict__ motion) {
  __shared__ uint64_t s_mask[kChunks];
  __shared__ uint32_t s_cbase[kChunks];
  __shared__ uint32_t s_tmp[32];

  const uint32_t b = blockIdx.x;
  const uint32_t n = min(n_per_scan[b], min(n_stride, kMaxN));  // never past the slot
  const uint2 *scan = nodes + (size_t)b * n_stride;
  float4 *out = xyzi + (size_t)b * out_stride;
  // E6 (include/rplgpu_msg.h): motion of the sensor during the scan, (vx, vy, wz, time_increment)
  // and, optionally, the time of the scan's first sample relative to the instant its points are wanted
  // at (rplgpu_set_scan_time_offsets_dev)
  float mvx = 0.0f, mvy = 0.0f, mwz = 0.0f, mdt = 0.0f, mt0 = 0.0f;
  const bool has_t0 = motion && T.scan_t0;  // block-uniform
  if (motion) {
    mvx = motion[4 * b], mvy = motion[4 * b + 1], mwz = motion[4 * b + 2], mdt = motion[4 * b + 3];
    if (has_t0) mt0 = T.scan_t0[b];
  }

  uint2 v[kIters];
  load_scan(scan, n, v);
  uint32_t kept = 0;
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    uint32_t d = nd_dist(v[j]);
    if (nd_keep(d, nd_quality(v[j]), p)) kept |= 1u << j;
  }
  if (keepmask) {  // E5: the radius-outlier mask replaces the E1 mask (it already includes it)
    const uint32_t *m = keepmask + (size_t)b * mask_stride;
#pragma unroll
    for (int j = 0; j < kIters; ++j) {
      const uint32_t word = 2u * ((uint32_t)(j * kWaves) + wave_id()) + (lane_id() >> 5);
      const uint32_t bits = (word < mask_stride) ? m[word] : 0u;
      if (!((bits >> (lane_id() & 31u)) & 1u)) kept &= ~(1u << j);
    }
  }
  const uint32_t count = publish_masks_and_scan(kept, s_mask, s_cbase, s_tmp);
  if (threadIdx.x == 0) {
    n_points[b] = min(count, out_stride);
    if (status) status[b] = (count > out_stride) ? RPLGPU_SCAN_OUT_TRUNCATED : 0u;
  }
  const float2 *cs = p.inverted ? T.cs_inv : T.cs;
#pragma unroll
  for (int j = 0; j < kIters; ++j) {
    if ((kept >> j) & 1u) {
      uint32_t r = sample_rank(j, s_mask, s_cbase);
      if (r < out_stride) {
        float dm = nd_dist_m(nd_dist(v[j]));
        float2 c = cs[nd_q14(v[j])];
        float x = dm * c.x, y = dm * c.y;
        if (motion) {  // the point as seen from the sensor pose at the first sample
          float tau = (float)sample_index(j) * mdt;
          if (has_t0) tau = mt0 + tau;
          const float a = mwz * tau, a2 = a * a;
          // sin / cos as fixed polynomials (|a| <= 0.5 rad: error < 2e-8), every operation
          // rounded once, in this order -- the oracle does the same, bit for bit
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
          const float x1 = (cn * x - sn * y) + mvx * tau;
          const float y1 = (sn * x + cn * y) + mvy * tau;
          x = x1, y = y1;
        }
        out[r] = make_float4(x, y, 0.0f, nd_intensity(nd_quality(v[j]), p.is_new_protocol));
      }
    }
  }
}

// ------------------------------------------------------------------------------
// pack: per-scan regions -> one contiguous cloud (scan order)
// ------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_offsets(const uint32_t *__restrict__ n_points,
                                                    uint32_t B, uint64_t *__restrict__ offsets) {
  __shared__ uint32_t s_tmp[32];
  __shared__ uint64_t s_carry;
  if (threadIdx.x == 0) s_carry = 0ull;
  __syncthreads();
  for (uint32_t base = 0; base < B; base += kBlock) {
    uint32_t i = base + threadIdx.x;
    uint32_t v = (i < B) ? n_points[i] : 0u;
    uint32_t total;
    uint32_t ex = block_excl_scan(v, s_tmp, &total);
    uint64_t carry = s_carry;
    if (i < B) offsets[i] = carry + ex;
    __syncthreads();
    if (threadIdx.x == 0) s_carry = carry + total;
    __syncthreads();
  }
  if (threadIdx.x == 0) offsets[B] = s_carry;
}

__global__ __launch_bounds__(256) void k_pack(const float4 *__restrict__ xyzi, uint32_t out_stride,
                                              const uint32_t *__restrict__ n_points,
                                              const uint64_t *__restrict__ offsets,
                                              float4 *__restrict__ packed) {
  const uint32_t b = blockIdx.x;
  const uint32_t m = n_points[b];
  const float4 *src = xyzi + (size_t)b * out_stride;
  float4 *dst = packed + offsets[b];
  for (uint32_t t = threadIdx.x; t < m; t += 256) dst[t] = src[t];
}

// ------------------------------------------------------------------------------
// host-side launchers (declared in rpl_launch.hpp)
// ------------------------------------------------------------------------------
// which first kernel a call takes: the streaming kernel, or (a handful of LONG scans) k_ascend<false>
static bool ascend_streams(uint32_t B, uint32_t n_stride) { return !(B <= 8u && n_stride > 8192u); }

hipError_t launch_ascend(hipStream_t s, void *nodes, uint32_t n_stride, const uint32_t *n_per_scan,
                         uint32_t B, uint32_t *status, uint32_t *need_sort, bool defer_sort,
                         uint32_t *sort_stat) {
  if (B == 0) return hipSuccess;
  // (the list of unsorted scans starts empty: cleared when the handle is created and, after a
  // single-scan call, by the sorting kernel itself; batches clear it here)
  if (B != 1u)
    if (hipError_t e = hipMemsetAsync(need_sort, 0, 4, s); e != hipSuccess) return e;
  // defer_sort (single-scan calls that can read the status word back): the scan is only marked
  // (kAscendUnsorted) and the caller launches launch_ascend_sort when it sees the mark — the
  // sorting kernel is a 1024-thread, 128 KiB workgroup that a sorted scan (the usual case)
  // launched for nothing, 2 us of the call
  const uint32_t mark = (defer_sort && B == 1u) ? kAscendUnsorted : 0u;
  // A handful of LONG scans (the single-scan seam above the SDK's 8192-node cap): latency counts,
  // and one 1024-thread workgroup that holds the scan in registers finishes a 32 000-sample scan in
  // a quarter less time than one 256-thread workgroup streaming it (46 vs 62 us per call; at 360
  // samples the streaming kernel is the faster one, 18 vs 24 us); batches stream.
  if (!ascend_streams(B, n_stride))
    hipLaunchKernelGGL(k_ascend<false>, dim3(B), dim3(kBlock), 0, s, (uint2 *)nodes, n_stride, n_per_scan,
                       status, need_sort, mark, 0u, (uint32_t *)nullptr);
  else
    hipLaunchKernelGGL(k_ascend_stream, dim3(B), dim3(kAscT), 0, s, (uint2 *)nodes, n_stride, n_per_scan,
                       status, need_sort, mark);
  if (mark) return hipGetLastError();
  return launch_ascend_sort(s, nodes, n_stride, n_per_scan, B, status, need_sort, sort_stat);
}

// the sorting kernel over the list the kernels above left (second half of launch_ascend)
hipError_t launch_ascend_sort(hipStream_t s, void *nodes, uint32_t n_stride, const uint32_t *n_per_scan,
                              uint32_t B, uint32_t *status, uint32_t *need_sort, uint32_t *sort_stat) {
  if (B == 0) return hipSuccess;
  // (scans queued by the streaming kernel have their filled angle words in place and may have been
  // permuted by its local repair: the sort then trusts the stored words)
  const uint32_t prefilled = ascend_streams(B, n_stride) ? 1u : 0u;
  hipLaunchKernelGGL(k_ascend<true>, dim3(std::min<uint32_t>(B, 256u)), dim3(kBlock), 0, s,
                     (uint2 *)nodes, n_stride, n_per_scan, status, need_sort, 0u, prefilled, sort_stat);
  if (B != 1u)  // (invariant between calls: the list is empty)
    if (hipError_t e = hipMemsetAsync(need_sort, 0, 4, s); e != hipSuccess) return e;
  return hipGetLastError();
}

hipError_t launch_laserscan_raw(hipStream_t s, const void *nodes, uint32_t n_stride,
                                const uint32_t *n_per_scan, uint32_t B, const KParams &p,
                                float *ranges, float *intens, uint32_t *beam_count) {
  if (B == 0) return hipSuccess;
  hipLaunchKernelGGL(k_laserscan_raw, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                     n_per_scan, p, ranges, intens, beam_count);
  return hipGetLastError();
}

hipError_t launch_cloud(hipStream_t s, const void *nodes, uint32_t n_stride,
                        const uint32_t *n_per_scan, uint32_t B, const KParams &p, const Tables &T,
                        bool voxel, const uint32_t *keepmask, uint32_t mask_stride, float *xyzi,
                        uint32_t out_stride, uint32_t *n_points, uint32_t *status,
                        const float *motion) {
  if (B == 0) return hipSuccess;
  if (voxel) {
    return launch_cloud_voxel(s, nodes, n_stride, n_per_scan, B, p, T, keepmask, mask_stride, xyzi,
                              out_stride, n_points, status);
  } else {
    hipLaunchKernelGGL(k_cloud, dim3(B), dim3(kBlock), 0, s, (const uint2 *)nodes, n_stride,
                       n_per_scan, p, T, keepmask, mask_stride, (float4 *)xyzi, out_stride,
                       n_points, status, motion);
  }
  return hipGetLastError();
}

hipError_t launch_pack(hipStream_t s, const float *xyzi, uint32_t out_stride,
                       const uint32_t *n_points, uint32_t B, float *packed, uint64_t *offsets) {
  hipLaunchKernelGGL(k_offsets, dim3(1), dim3(kBlock), 0, s, n_points, B, offsets);
  if (B)
    hipLaunchKernelGGL(k_pack, dim3(B), dim3(256), 0, s, (const float4 *)xyzi, out_stride,
                       n_points, offsets, (float4 *)packed);
  return hipGetLastError();
}

}  // namespace rpl
