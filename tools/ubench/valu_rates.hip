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
  return 0;
}
