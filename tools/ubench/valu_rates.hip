// Developer aid: per-SIMD issue cost (cycles per wave-instruction) of the VALU / DPP / LDS
// instructions the scan kernels are built from, on gfx950.  Each kernel runs 8 independent
// dependency chains of ONE instruction; waves/SIMD is swept so both latency-bound and
// throughput-bound regimes show.   hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define ITERS 2048
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFK(NAME, ASMSTR)                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(float *out, float s0, float s1) {     \
    float v0 = threadIdx.x * s0, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4,     \
          v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;                                         \
    float a = s0, b = s1;                                                                \
    for (int i = 0; i < ITERS; ++i) {                                                    \
      asm volatile(ASMSTR : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5),  \
                   "+v"(v6), "+v"(v7) : "v"(a), "v"(b));                                \
    }                                                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;  \
  }

#define I8(OP, TAIL) OP " %0, %0" TAIL "\n" OP " %1, %1" TAIL "\n" OP " %2, %2" TAIL "\n" OP " %3, %3" TAIL "\n" \
                     OP " %4, %4" TAIL "\n" OP " %5, %5" TAIL "\n" OP " %6, %6" TAIL "\n" OP " %7, %7" TAIL "\n"
#define I8x2(OP, TAIL) I8(OP, TAIL) I8(OP, TAIL)

DEFK(fma, I8x2("v_fma_f32", ", %8, %9"))
DEFK(mul, I8x2("v_mul_f32", ", %8"))
DEFK(add, I8x2("v_add_f32", ", %8"))
DEFK(addu, I8x2("v_add_u32", ", %8"))
DEFK(cvt_f32_u32, I8x2("v_cvt_f32_u32", ""))
DEFK(cvt_i32_f32, I8x2("v_cvt_i32_f32", ""))
DEFK(floor, I8x2("v_floor_f32", ""))
DEFK(rndne, I8x2("v_rndne_f32", ""))
DEFK(rcp, I8x2("v_rcp_f32", ""))
DEFK(alignbit, I8x2("v_alignbit_b32", ", %8, 16"))
DEFK(bfe, I8x2("v_bfe_u32", ", 16, 8"))
DEFK(lshl_or, I8x2("v_lshl_or_b32", ", 16, %8"))
DEFK(mul_lo, I8x2("v_mul_lo_u32", ", %8"))
DEFK(mad_u32_u24, I8x2("v_mad_u32_u24", ", %8, %9"))
DEFK(med3, I8x2("v_med3_i32", ", %8, %9"))
DEFK(cndmask, I8x2("v_cndmask_b32", ", %8, vcc"))
DEFK(dpp_shr1, I8x2("v_add_u32_dpp", ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"))
DEFK(dpp_wave_shr, I8x2("v_mov_b32_dpp", " wave_shr:1 row_mask:0xf bank_mask:0xf"))
DEFK(dpp_bcast15, I8x2("v_add_u32_dpp", ", %8 row_bcast:15 row_mask:0xa bank_mask:0xf"))
DEFK(cmp, "v_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\nv_cmp_lt_f32 vcc, %3, %8\n"
          "v_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\nv_cmp_lt_f32 vcc, %7, %8\n"
          "v_cmp_lt_f32 vcc, %0, %9\nv_cmp_lt_f32 vcc, %1, %9\nv_cmp_lt_f32 vcc, %2, %9\nv_cmp_lt_f32 vcc, %3, %9\n"
          "v_cmp_lt_f32 vcc, %4, %9\nv_cmp_lt_f32 vcc, %5, %9\nv_cmp_lt_f32 vcc, %6, %9\nv_cmp_lt_f32 vcc, %7, %9\n")

// packed fp32: operands are register pairs
#define DEFK2(NAME, OP, TAIL)                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(float *out, float s0, float s1) {        \
    typedef float f2 __attribute__((ext_vector_type(2)));                                  \
    f2 v0 = {threadIdx.x * s0, 1.f}, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f,           \
       v4 = v0 + 4.f, v5 = v0 + 5.f, v6 = v0 + 6.f, v7 = v0 + 7.f;                          \
    f2 a = {s0, s1}, b = {s1, s0};                                                          \
    for (int i = 0; i < ITERS; ++i) {                                                       \
      asm volatile(I8x2(OP, TAIL) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4),       \
                   "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));                         \
    }                                                                                       \
    f2 s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;                                           \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;                                 \
  }
DEFK2(pk_fma, "v_pk_fma_f32", ", %8, %9")
DEFK2(pk_mul, "v_pk_mul_f32", ", %8")
DEFK2(pk_add, "v_pk_add_f32", ", %8")

#define DEFK3(NAME, OP, TAIL)                                                              \
  __global__ __launch_bounds__(256) void k_##NAME(float *out, float s0, float s1) {        \
    double v0 = threadIdx.x * s0, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4,       \
           v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;                                           \
    double a = s0, b = s1;                                                                  \
    for (int i = 0; i < ITERS; ++i) {                                                       \
      asm volatile(I8x2(OP, TAIL) : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4),       \
                   "+v"(v5), "+v"(v6), "+v"(v7) : "v"(a), "v"(b));                         \
    }                                                                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7); \
  }
DEFK3(fma_f64, "v_fma_f64", ", %8, %9")
DEFK3(rcp_f64, "v_rcp_f64", "")

// LDS: bpermute and b128 read / write
__global__ __launch_bounds__(256) void k_bpermute(float *out, float s0, float s1) {
  int v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x + k;
  int addr = ((threadIdx.x * 7) & 63) << 2;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_ds_bpermute(addr, v[k]);
  }
  int s = 0;
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_ds_write_b128(float *out, float s0, float s1) {
  __shared__ uint4 buf[256 * 8];
  uint4 v = make_uint4(threadIdx.x, 1, 2, 3);
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      buf[threadIdx.x + 256 * (k & 7)] = v;
      asm volatile("" ::: "memory");
    }
  }
  __syncthreads();
  out[blockIdx.x * blockDim.x + threadIdx.x] = buf[(threadIdx.x * 5) & 2047].x;
}
__global__ __launch_bounds__(256) void k_ds_read_b128(float *out, float s0, float s1) {
  __shared__ uint4 buf[256 * 8];
  for (int k = 0; k < 8; ++k) buf[threadIdx.x + 256 * k] = make_uint4(threadIdx.x, k, 2, 3);
  __syncthreads();
  uint32_t acc = 0;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint4 t;
      asm volatile("ds_read_b128 %0, %1\ns_waitcnt lgkmcnt(0)" : "=v"(t) : "v"((uint32_t)((threadIdx.x + 256 * (k & 7)) * 16)) : "memory");
      acc += t.x;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}


DEFK(and_b32, I8x2("v_and_b32", ", %8"))
DEFK(or_b32, I8x2("v_or_b32", ", %8"))
DEFK(lshlrev, I8x2("v_lshlrev_b32", ", 3"))
DEFK(lshrrev, "v_lshrrev_b32 %0, 3, %0\nv_lshrrev_b32 %1, 3, %1\nv_lshrrev_b32 %2, 3, %2\nv_lshrrev_b32 %3, 3, %3\nv_lshrrev_b32 %4, 3, %4\nv_lshrrev_b32 %5, 3, %5\nv_lshrrev_b32 %6, 3, %6\nv_lshrrev_b32 %7, 3, %7\n"
              "v_lshrrev_b32 %0, 3, %0\nv_lshrrev_b32 %1, 3, %1\nv_lshrrev_b32 %2, 3, %2\nv_lshrrev_b32 %3, 3, %3\nv_lshrrev_b32 %4, 3, %4\nv_lshrrev_b32 %5, 3, %5\nv_lshrrev_b32 %6, 3, %6\nv_lshrrev_b32 %7, 3, %7\n")
DEFK(sub_u32, I8x2("v_sub_u32", ", %8"))
DEFK(sub_f32, I8x2("v_sub_f32", ", %8"))
DEFK(max_f32, I8x2("v_max_f32", ", %8"))
DEFK(min_u32, I8x2("v_min_u32", ", %8"))
DEFK(mov, "v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %8\n"
          "v_mov_b32 %0, %1\nv_mov_b32 %1, %2\nv_mov_b32 %2, %3\nv_mov_b32 %3, %4\nv_mov_b32 %4, %5\nv_mov_b32 %5, %6\nv_mov_b32 %6, %7\nv_mov_b32 %7, %9\n")
DEFK(cvt_ubyte2, I8x2("v_cvt_f32_ubyte2", ""))
DEFK(perm, I8x2("v_perm_b32", ", %8, %9"))
DEFK(pack, I8x2("v_pack_b32_f16", ", %8"))
DEFK(fmac, I8x2("v_fmac_f32", ", %8"))
DEFK(cvt_flr, I8x2("v_cvt_flr_i32_f32", ""))
DEFK(fract, I8x2("v_fract_f32", ""))
DEFK(add3, I8x2("v_add3_u32", ", %8, %9"))
DEFK(lshl_add, I8x2("v_lshl_add_u32", ", 3, %8"))
DEFK(and_or, I8x2("v_and_or_b32", ", %8, %9"))
DEFK(mbcnt, I8x2("v_mbcnt_lo_u32_b32", ", -1"))
DEFK(sdwa_lshl, I8x2("v_lshlrev_b32_sdwa", ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"))
DEFK(cndmask64, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\n"
                "v_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]\n"
                "v_cndmask_b32_e64 %0, %0, %9, s[20:21]\nv_cndmask_b32_e64 %1, %1, %9, s[20:21]\nv_cndmask_b32_e64 %2, %2, %9, s[20:21]\nv_cndmask_b32_e64 %3, %3, %9, s[20:21]\n"
                "v_cndmask_b32_e64 %4, %4, %9, s[20:21]\nv_cndmask_b32_e64 %5, %5, %9, s[20:21]\nv_cndmask_b32_e64 %6, %6, %9, s[20:21]\nv_cndmask_b32_e64 %7, %7, %9, s[20:21]\n")
DEFK(cmp_u32_sgpr, "v_cmp_lt_u32_e64 s[20:21], %0, %8\nv_cmp_lt_u32_e64 s[22:23], %1, %8\nv_cmp_lt_u32_e64 s[20:21], %2, %8\nv_cmp_lt_u32_e64 s[22:23], %3, %8\n"
                   "v_cmp_lt_u32_e64 s[20:21], %4, %8\nv_cmp_lt_u32_e64 s[22:23], %5, %8\nv_cmp_lt_u32_e64 s[20:21], %6, %8\nv_cmp_lt_u32_e64 s[22:23], %7, %8\n"
                   "v_cmp_lt_u32_e64 s[20:21], %0, %9\nv_cmp_lt_u32_e64 s[22:23], %1, %9\nv_cmp_lt_u32_e64 s[20:21], %2, %9\nv_cmp_lt_u32_e64 s[22:23], %3, %9\n"
                   "v_cmp_lt_u32_e64 s[20:21], %4, %9\nv_cmp_lt_u32_e64 s[22:23], %5, %9\nv_cmp_lt_u32_e64 s[20:21], %6, %9\nv_cmp_lt_u32_e64 s[22:23], %7, %9\n")

// gather: dwordx2 loads from a 512 KiB table, lane stride `stride` entries (8 B each)
__global__ __launch_bounds__(256) void k_gather(const uint2 *__restrict__ tab, float *out, uint32_t stride, uint32_t step) {
  uint32_t idx = (blockIdx.x * 977u + (threadIdx.x >> 6) * 4099u + (threadIdx.x & 63) * stride) & 65535u;
  uint32_t acc = 0;
  for (int i = 0; i < ITERS / 4; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      uint2 v = tab[idx];
      acc += v.x ^ v.y;
      idx = (idx + step) & 65535u;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
// 16-byte loads with a per-lane stride (bytes), streaming over a large buffer
__global__ __launch_bounds__(256) void k_strided16(const uint4 *__restrict__ buf, float *out, uint32_t lane_stride16, uint32_t nvec) {
  // each wave owns a contiguous region of 64*lane_stride16 vectors; lanes walk their own sub-range
  uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t lane = threadIdx.x & 63;
  uint32_t acc = 0;
  size_t base = (size_t)wave * 64 * lane_stride16 * 8;  // 8 regions per wave
  for (int r = 0; r < 8; ++r) {
    const uint4 *p = buf + (base + (size_t)r * 64 * lane_stride16 + (size_t)lane * lane_stride16) % nvec;
    for (uint32_t j = 0; j < lane_stride16; ++j) { uint4 v = p[j]; acc += v.x ^ v.w; }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

typedef void (*kfn)(float *, float, float);
struct Ent { const char *name; kfn fn; int per_iter; };

int main() {
  float *d; hipMalloc(&d, 256 * 64 * 256 * 4 * 8);
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  double ghz = prop.clockRate / 1e6;
  printf("device %s CUs %d clock %.2f GHz\n", prop.gcnArchName, cus, ghz);
  std::vector<Ent> ents = {
    {"v_fma_f32", k_fma, 16}, {"v_mul_f32", k_mul, 16}, {"v_add_f32", k_add, 16}, {"v_add_u32", k_addu, 16},
    {"v_pk_fma_f32", k_pk_fma, 16}, {"v_pk_mul_f32", k_pk_mul, 16}, {"v_pk_add_f32", k_pk_add, 16},
    {"v_cvt_f32_u32", k_cvt_f32_u32, 16}, {"v_cvt_i32_f32", k_cvt_i32_f32, 16}, {"v_floor_f32", k_floor, 16},
    {"v_rndne_f32", k_rndne, 16}, {"v_rcp_f32", k_rcp, 16}, {"v_alignbit_b32", k_alignbit, 16},
    {"v_bfe_u32", k_bfe, 16}, {"v_lshl_or_b32", k_lshl_or, 16}, {"v_mul_lo_u32", k_mul_lo, 16},
    {"v_mad_u32_u24", k_mad_u32_u24, 16}, {"v_med3_i32", k_med3, 16}, {"v_cndmask_b32", k_cndmask, 16},
    {"v_cmp_lt_f32", k_cmp, 16}, {"dpp row_shr:1 add", k_dpp_shr1, 16}, {"dpp wave_shr:1 mov", k_dpp_wave_shr, 16},
    {"dpp row_bcast:15 add", k_dpp_bcast15, 16}, {"v_fma_f64", k_fma_f64, 16}, {"v_rcp_f64", k_rcp_f64, 16},
    {"v_and_b32", k_and_b32, 16}, {"v_or_b32", k_or_b32, 16}, {"v_lshlrev_b32", k_lshlrev, 16}, {"v_lshrrev_b32", k_lshrrev, 16},
    {"v_sub_u32", k_sub_u32, 16}, {"v_sub_f32", k_sub_f32, 16}, {"v_max_f32", k_max_f32, 16}, {"v_min_u32", k_min_u32, 16},
    {"v_mov_b32", k_mov, 16}, {"v_cvt_f32_ubyte2", k_cvt_ubyte2, 16}, {"v_perm_b32", k_perm, 16}, {"v_pack_b32_f16", k_pack, 16},
    {"v_fmac_f32", k_fmac, 16}, {"v_cvt_flr_i32_f32", k_cvt_flr, 16}, {"v_fract_f32", k_fract, 16}, {"v_add3_u32", k_add3, 16},
    {"v_lshl_add_u32", k_lshl_add, 16}, {"v_and_or_b32", k_and_or, 16}, {"v_mbcnt_lo", k_mbcnt, 16}, {"v_lshlrev_b32_sdwa", k_sdwa_lshl, 16},
    {"v_cndmask_b32_e64 sgpr", k_cndmask64, 16}, {"v_cmp_lt_u32_e64 sgpr", k_cmp_u32_sgpr, 16},
    {"ds_bpermute_b32", k_bpermute, 16}, {"ds_write_b128", k_ds_write_b128, 16}, {"ds_read_b128", k_ds_read_b128, 16},
  };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  printf("%-22s", "cycles/wave-instr/SIMD @ waves/SIMD:");
  for (int w : {1, 2, 4, 8}) printf(" %7d", w);
  printf("\n");
  for (auto &e : ents) {
    printf("%-36s", e.name);
    for (int w : {1, 2, 4, 8}) {
      int blocks = cus * w;  // 256 threads = 4 waves = one per SIMD
      hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
      hipDeviceSynchronize();
      float best = 1e9;
      for (int r = 0; r < 3; ++r) {
        hipEventRecord(a);
        hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), 0, 0, d, 1.0f, 0.5f);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
      }
      double instr_per_simd = (double)ITERS * e.per_iter * w;
      printf(" %7.2f", best * 1e-3 * ghz * 1e9 / instr_per_simd);
    }
    printf("\n");
  }
  // ---- gather cost vs lane stride
  uint2 *tab; hipMalloc(&tab, 65536 * 8); hipMemset(tab, 1, 65536 * 8);
  printf("gather dwordx2 from 512 KiB table: ns per wave-instr per CU (8 waves/SIMD) by lane stride (entries)\n");
  for (uint32_t stride : {1u, 2u, 4u, 8u, 16u, 32u, 64u, 1021u}) {
    int w = 8; int blocks = cus * w;
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, d, stride, 2u);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, d, stride, 2u);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_cu = (double)(ITERS / 4) * 8 * w * 4;
    printf("  stride %5u: %.2f cycles/wave-instr/CU\n", stride, ms * 1e-3 * ghz * 1e9 / instr_per_cu);
  }
  // ---- strided 16-byte streaming loads
  size_t nvec = (size_t)1 << 26;  // 1 GiB
  uint4 *big; hipMalloc(&big, nvec * 16); hipMemset(big, 1, nvec * 16);
  printf("16-B loads, lane stride S*16 B (each lane walks S vectors): effective GB/s\n");
  for (uint32_t S : {1u, 2u, 4u, 8u, 16u}) {
    int blocks = (int)(nvec / (4 * 8 * 64 * (size_t)S));
    hipLaunchKernelGGL(k_strided16, dim3(blocks), dim3(256), 0, 0, big, d, S, (uint32_t)nvec);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k_strided16, dim3(blocks), dim3(256), 0, 0, big, d, S, (uint32_t)nvec);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    printf("  S=%2u: %.0f GB/s (%.3f ms)\n", S, nvec * 16.0 / (ms * 1e-3) / 1e9, ms);
  }
  return 0;
}
