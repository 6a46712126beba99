// Developer aid: how fast can ONE compute unit stream HBM, as a function of the loads a wave keeps
// in flight?  One 1024-thread workgroup per CU (100 KiB of LDS pins it there), every workgroup
// reads its own contiguous 256 KB "scans" one after the other with 16-byte loads per lane, DEPTH
// independent loads issued back to back before the first one is consumed.
//   hipcc --offload-arch=gfx950 -O3 cu_stream.hip -o cu_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int DEPTH>
__global__ __launch_bounds__(1024) void k_stream(const uint4 *__restrict__ src, size_t per_wg_vec,
                                                 uint32_t *out) {
  __shared__ uint32_t pin[25 * 1024];
  pin[threadIdx.x] = threadIdx.x;
  const uint4 *p = src + (size_t)blockIdx.x * per_wg_vec;
  uint32_t acc = 0;
  uint4 buf[DEPTH];
  size_t i = threadIdx.x;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = p[i + (size_t)d * 1024];
  for (; i + (size_t)DEPTH * 1024 < per_wg_vec; i += (size_t)DEPTH * 1024) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const uint4 v = buf[d];
      buf[d] = p[i + (size_t)(DEPTH + d) * 1024];  // next block of this slot
      acc += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  __syncthreads();
  out[blockIdx.x * 1024 + threadIdx.x] = acc + pin[(threadIdx.x * 7) & 1023];
}

int main() {
  const size_t total = 1ull << 30;  // 1 GiB
  uint4 *src;
  uint32_t *out;
  hipMalloc(&src, total + (1 << 24));
  hipMemset(src, 1, total + (1 << 24));
  hipMalloc(&out, 256 * 1024 * 4);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int grids[] = {8, 64, 256};
  printf("GB/s per CU (and total) by loads in flight per wave; 1024-thread workgroup per CU\n");
  for (int g : grids) {
    const size_t per_wg = (total / g / 16) & ~size_t(1023 * 16 + 15);
    auto run = [&](auto kern, int depth) {
      float best = 1e9f;
      for (int it = 0; it < 3; ++it) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(g), dim3(1024), 0, 0, src, per_wg, out);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
      }
      const double gb = (double)per_wg * 16 * g / 1e9;
      printf("  grid %3d depth %d: %7.1f GB/s per CU, %7.1f GB/s total (%.3f ms)\n", g, depth,
             gb / (best * 1e-3) / g, gb / (best * 1e-3), best);
    };
    run(k_stream<1>, 1);
    run(k_stream<2>, 2);
    run(k_stream<3>, 3);
    run(k_stream<4>, 4);
    run(k_stream<6>, 6);
    run(k_stream<8>, 8);
  }
  return 0;
}
