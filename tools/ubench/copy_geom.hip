// Developer aid: what does the SHAPE of a copy cost on MI355X?  0.5 GB is copied (or only read)
//   A  "one workgroup per scan":   workgroup g copies its own contiguous CHUNK (128 KB) from
//      src + g*src_stride to dst + g*dst_stride — thousands of independent sequential streams;
//   B  "pieces":                   the same chunks cut into 16 KB pieces, piece index fastest in
//      the grid — neighbouring workgroups work on neighbouring memory;
//   C  flat grid-stride copy of one contiguous 0.5 GB (the ideal).
//   hipcc --offload-arch=gfx950 -O3 copy_geom.hip -o copy_geom
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

constexpr int kT = 256;

template <bool WRITE>
__global__ __launch_bounds__(kT) void k_chunks(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                               size_t src_stride, size_t dst_stride, uint32_t n4,
                                               uint32_t pieces, uint32_t *sink) {
  // blockIdx.x = piece, blockIdx.y = chunk; a piece is n4 / pieces vectors
  const uint32_t per = n4 / pieces;
  const uint4 *s = src + (size_t)blockIdx.y * src_stride + (size_t)blockIdx.x * per;
  uint4 *d = dst + (size_t)blockIdx.y * dst_stride + (size_t)blockIdx.x * per;
  uint32_t acc = 0;
  uint32_t t = threadIdx.x;
  for (; t + 3u * kT < per; t += 4u * kT) {
    const uint4 a = s[t], b = s[t + kT], c = s[t + 2 * kT], e = s[t + 3 * kT];
    if (WRITE) {
      d[t] = a; d[t + kT] = b; d[t + 2 * kT] = c; d[t + 3 * kT] = e;
    } else {
      acc += a.x ^ b.y ^ c.z ^ e.w;
    }
  }
  for (; t < per; t += kT) {
    const uint4 a = s[t];
    if (WRITE) d[t] = a; else acc += a.x;
  }
  if (!WRITE && acc == 0x12345678u) sink[0] = acc;
}

template <bool WRITE>
__global__ __launch_bounds__(kT) void k_flat(const uint4 *__restrict__ src, uint4 *__restrict__ dst,
                                             size_t n4, uint32_t *sink) {
  uint32_t acc = 0;
  for (size_t t = (size_t)blockIdx.x * kT + threadIdx.x; t < n4; t += (size_t)gridDim.x * kT) {
    const uint4 a = src[t];
    if (WRITE) dst[t] = a; else acc += a.x;
  }
  if (!WRITE && acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(kT) void k_fill(uint4 *__restrict__ dst, size_t n4, uint32_t seed) {
  for (size_t t = (size_t)blockIdx.x * kT + threadIdx.x; t < n4; t += (size_t)gridDim.x * kT)
    dst[t] = make_uint4(seed, (uint32_t)t, seed ^ (uint32_t)t, 7u);
}
// one workgroup per 256 KB chunk, every lane writes 32 B per trip as two 16-byte halves (the
// decode kernel's store pattern), or as one coalesced 16-byte store per lane and instruction
template <bool HALVES>
__global__ __launch_bounds__(kT) void k_fill_chunks(uint4 *__restrict__ dst, size_t stride4, uint32_t n4,
                                                    uint32_t seed) {
  uint4 *d = dst + (size_t)blockIdx.x * stride4;
  if (HALVES) {
    for (uint32_t t = threadIdx.x; 2u * t + 1u < n4; t += kT) {
      d[2u * t] = make_uint4(seed, t, 1u, 2u);
      d[2u * t + 1u] = make_uint4(seed, t, 3u, 4u);
    }
  } else {
    for (uint32_t t = threadIdx.x; t < n4; t += kT) d[t] = make_uint4(seed, t, 1u, 2u);
  }
}

int main() {
  const uint32_t chunks = 4096, chunk_bytes = 128 * 1024, n4 = chunk_bytes / 16;
  const size_t src_stride4 = 256320 / 16, dst_stride4 = 4 * 256320 / 16;  // the decode bench's strides
  uint4 *src, *dst;
  uint32_t *sink;
  hipMalloc(&src, (size_t)chunks * src_stride4 * 16 + (1 << 20));
  hipMalloc(&dst, (size_t)chunks * dst_stride4 * 16 + (1 << 20));
  hipMalloc(&sink, 64);
  hipMemset(src, 1, (size_t)chunks * src_stride4 * 16);
  hipMemset(dst, 0, (size_t)chunks * dst_stride4 * 16);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const double gb = (double)chunks * chunk_bytes / 1e9;
  auto time = [&](const char *what, auto launch, double traffic_gb) {
    float best = 1e9f;
    for (int it = 0; it < 4; ++it) {
      hipEventRecord(a);
      launch();
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = ms < best ? ms : best;
    }
    printf("%-58s %.3f ms  %7.0f GB/s\n", what, best, traffic_gb / (best * 1e-3));
  };
  for (uint32_t pieces : {1u, 2u, 4u, 8u, 16u}) {
    char name[128];
    snprintf(name, sizeof name, "A/B copy  %u chunks x %u pieces (strided src/dst)", chunks, pieces);
    time(name, [&] { hipLaunchKernelGGL(k_chunks<true>, dim3(pieces, chunks), dim3(kT), 0, 0, src, dst, src_stride4, dst_stride4, n4, pieces, sink); }, 2 * gb);
    snprintf(name, sizeof name, "A/B read  %u chunks x %u pieces (strided src)", chunks, pieces);
    time(name, [&] { hipLaunchKernelGGL(k_chunks<false>, dim3(pieces, chunks), dim3(kT), 0, 0, src, dst, src_stride4, dst_stride4, n4, pieces, sink); }, gb);
  }
  // dense (unstrided) chunks: the same 4096 streams, back to back in memory
  time("A copy  4096 chunks x 1 piece, chunks back to back", [&] { hipLaunchKernelGGL(k_chunks<true>, dim3(1, chunks), dim3(kT), 0, 0, src, dst, (size_t)n4, (size_t)n4, n4, 1u, sink); }, 2 * gb);
  time("A read  4096 chunks x 1 piece, chunks back to back", [&] { hipLaunchKernelGGL(k_chunks<false>, dim3(1, chunks), dim3(kT), 0, 0, src, dst, (size_t)n4, (size_t)n4, n4, 1u, sink); }, gb);
  for (uint32_t g : {1024u, 4096u, 16384u}) {
    char name[128];
    snprintf(name, sizeof name, "C flat copy, grid %u", g);
    time(name, [&] { hipLaunchKernelGGL(k_flat<true>, dim3(g), dim3(kT), 0, 0, src, dst, (size_t)chunks * n4, sink); }, 2 * gb);
    snprintf(name, sizeof name, "C flat read, grid %u", g);
    time(name, [&] { hipLaunchKernelGGL(k_flat<false>, dim3(g), dim3(kT), 0, 0, src, dst, (size_t)chunks * n4, sink); }, gb);
  }
  // write-only: 1 GiB
  const size_t fill4 = (1ull << 30) / 16;
  for (uint32_t g : {1024u, 4096u, 16384u}) {
    char name[128];
    snprintf(name, sizeof name, "W flat fill 1 GiB, grid %u", g);
    time(name, [&] { hipLaunchKernelGGL(k_fill, dim3(g), dim3(kT), 0, 0, dst, fill4, 5u); }, 1.0737);
  }
  time("W 4096 chunks x 256 KB, 16 B per lane and store", [&] { hipLaunchKernelGGL(k_fill_chunks<false>, dim3(4096), dim3(kT), 0, 0, dst, (size_t)256320 / 16, 16384u, 5u); }, 1.0737);
  time("W 4096 chunks x 256 KB, 32 B per lane as two halves", [&] { hipLaunchKernelGGL(k_fill_chunks<true>, dim3(4096), dim3(kT), 0, 0, dst, (size_t)256320 / 16, 16384u, 5u); }, 1.0737);
  hipMemsetAsync(dst, 0, 1ull << 30, 0);
  time("W hipMemsetAsync 1 GiB", [&] { hipMemsetAsync(dst, 1, 1ull << 30, 0); }, 1.0737);
  return 0;
}
