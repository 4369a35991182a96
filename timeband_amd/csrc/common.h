// Shared helpers for the gfx950 kernels of the Timeband LSTM-GAN hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define TB_OK 0
#define TB_ERR_ARG 1
#define TB_ERR_HIP 2
#define TB_ERR_UNSUPPORTED 3
#define TB_ERR_WORKSPACE 4

// error reporting (capi.hip)
int tb_set_error(int code, const char* fmt, ...);
int tb_check_hip(hipError_t e, const char* what);
int tb_num_cus();

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
