// Shared helpers for the gfx950 kernels of the Timeband LSTM-GAN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define TB_OK 0
#define TB_ERR_ARG 1
#define TB_ERR_HIP 2
#define TB_ERR_UNSUPPORTED 3
#define TB_ERR_WORKSPACE 4

// error reporting (capi.hip)
int tb_set_error(int code, const char* fmt, ...);
int tb_check_hip(hipError_t e, const char* what);
int tb_num_cus();
size_t tb_gemm_ws_bytes();   // gemm.hip: upper bound of the split-K slabs
// A product whose split-K reduction (and / or `C +=`) is left to the caller: C (+)= sum over z < splits of
// slab[z][M][N], in that order.  splits == 0: nothing pending (the kernel wrote C itself).
struct TbGemmDeferred {
  const float* slab;
  int splits, M, N, accumulate;
  float* C;
  long ldc;
};
// tb_gemm_launch without its reduction kernels (no bias / activation); falls back to the complete product
// (d->splits = 0) where the operands have to be cut (> 2 GiB).
int tb_gemm_launch_deferred(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C, long ldc,
                            int M, int N, int K, int b_shift, int b_period, float* ws, size_t ws_bytes, hipStream_t stream,
                            int accumulate, TbGemmDeferred* d);

// losses.hip: column-sum partial pass and the one-launch tail of a linear layer's backward (capi.hip: tb_linear_bwd)
int tb_colsum_partials(const float* a, long rows, int cols, void* ws, size_t ws_bytes, hipStream_t s, int* nparts);
int tb_linear_finish(const TbGemmDeferred* d, const double* part, int nparts, int cols, float* db, int accumulate, hipStream_t s);

#define TB_HIP(call)                                   \
  do {                                                 \
    int _rc = tb_check_hip((call), #call);             \
    if (_rc) return _rc;                               \
  } while (0)

#define TB_LAUNCH_CHECK(name)                          \
  do {                                                 \
    int _rc = tb_check_hip(hipGetLastError(), name);   \
    if (_rc) return _rc;                               \
  } while (0)

static inline size_t tb_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- device helpers --------------------------------------------------------------------
__device__ __forceinline__ float tb_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
// tanh of a layer output (Linear + Tanh): 1 - 2 / (1 + e^(2x)) on the hardware exp and reciprocal -- absolute error ~2e-7 on
// values in (-1, 1), six instructions where tanhf is a branchy ~30.  ONE definition for every kernel that applies the
// activation (GEMM epilogues, the split-K finish, the fused output-layer + band kernel), so they agree bit for bit.
__device__ __forceinline__ float tb_tanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// Agent-scope (write-through / L1-bypassing) accesses used for the in-launch hand-offs
// between the workgroups of one cluster (guide: cdna_hip_programming.md Guideline 16).
__device__ __forceinline__ void tb_store_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float tb_load_agent(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tb_store_agent_u32(unsigned* p, unsigned v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void tb_store_agent_u64(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned tb_load_agent_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long tb_load_agent_u64(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- split-precision bf16 operands ("bf16x3") ------------------------------------------------
// An fp32 value x is carried as two bf16 values hi = bf16(x), lo = bf16(x - hi) (x = hi + lo to
// ~2^-17 relative).  A product a.b is then three bf16 MFMAs with fp32 accumulation,
// a_hi.b_hi + a_hi.b_lo + a_lo.b_hi (the dropped lo.lo term is ~2^-18 relative): fp32-grade results
// at 16/3 of the fp32 MFMA rate (v_mfma_f32_16x16x32_bf16 is 16x the rate of v_mfma_f32_16x16x4_f32).
__device__ __forceinline__ unsigned tb_cvt_pk_bf16(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0,%1,%2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
// two fp32 values -> packed bf16 pair of the high parts and packed pair of the residuals
__device__ __forceinline__ void tb_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = tb_cvt_pk_bf16(a, b);
  lo = tb_cvt_pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}
__device__ __forceinline__ void tb_split4(f32x4 v, u32x2& hi, u32x2& lo) {
  unsigned h0, l0, h1, l1;
  tb_split2(v[0], v[1], h0, l0);
  tb_split2(v[2], v[3], h1, l1);
  hi = (u32x2){h0, h1};
  lo = (u32x2){l0, l1};
}
__device__ __forceinline__ void tb_split8(f32x4 a, f32x4 b, bf16x8& hi, bf16x8& lo) {
  u32x2 h0, l0, h1, l1;
  tb_split4(a, h0, l0);
  tb_split4(b, h1, l1);
  hi = __builtin_bit_cast(bf16x8, (u32x4){h0[0], h0[1], h1[0], h1[1]});
  lo = __builtin_bit_cast(bf16x8, (u32x4){l0[0], l0[1], l1[0], l1[1]});
}
// acc += a.b with both operands split (small terms first)
__device__ __forceinline__ f32x4 tb_mfma_bf16x3(bf16x8 ah, bf16x8 al, bf16x8 bh, bf16x8 bl, f32x4 acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, acc, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, acc, 0, 0, 0);
}

// profiling hooks (capi.hip): HIP-event timing of one kernel family on its own stream
void tb_prof_begin(int kernel_id, hipStream_t s, double algo_bytes = 0.0);
void tb_prof_end(int kernel_id, hipStream_t s);

enum TbKernelId {
  TB_K_GEMM = 0,
  TB_K_LSTM_FWD = 1,
  TB_K_LSTM_BWD = 2,
  TB_K_LOSS = 3,
  TB_K_OPTIM = 4,
  TB_K_GATHER = 5,
  TB_K_BAND = 6,
  TB_K_COUNT = 8
};
