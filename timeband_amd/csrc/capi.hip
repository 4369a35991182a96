// C-ABI glue: error reporting, device info, kernel timing hooks, and the Linear entry points
// built on the K1 GEMM.  Declarations with reference citations: include/timeband_hip.h.
#include "common.h"

#include <stdarg.h>
#include <stdio.h>
#include <string.h>

int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream, int accumulate);

namespace {
thread_local char g_err[512] = "";
int g_cus = 0;

struct Prof {
  bool on = false;
  static constexpr int MAXE = 4096;
  hipEvent_t beg[MAXE], end[MAXE];
  int n = 0, created = 0;
  bool open = false;
  double bytes = 0.0;
};
Prof g_prof[TB_K_COUNT];
}  // namespace

int tb_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int tb_check_hip(hipError_t e, const char* what) {
  if (e == hipSuccess) return TB_OK;
  return tb_set_error(TB_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int tb_num_cus() {
  if (g_cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      g_cus = prop.multiProcessorCount;
    if (g_cus <= 0) g_cus = 256;
  }
  return g_cus;
}

void tb_prof_begin(int id, hipStream_t s, double algo_bytes) {
  Prof& p = g_prof[id];
  if (!p.on || p.n >= Prof::MAXE) return;
  if (p.n >= p.created) {
    if (hipEventCreate(&p.beg[p.n]) != hipSuccess || hipEventCreate(&p.end[p.n]) != hipSuccess) return;
    p.created = p.n + 1;
  }
  (void)hipEventRecord(p.beg[p.n], s);
  p.open = true;
  p.bytes += algo_bytes;
}

void tb_prof_end(int id, hipStream_t s) {
  Prof& p = g_prof[id];
  if (!p.on || !p.open) return;
  (void)hipEventRecord(p.end[p.n], s);
  p.n++;
  p.open = false;
}

extern "C" int tb_lstm_static_init(void);

extern "C" {

const char* tb_last_error(void) { return g_err; }

int tb_version(void) { return 1; }

int tb_init(int device) {
  TB_HIP(hipSetDevice(device));
  {
    const int rc = tb_lstm_static_init();   // lstm.hip: address of the static scratch on this device
    if (rc) return rc;
  }
  hipDeviceProp_t prop;
  TB_HIP(hipGetDeviceProperties(&prop, device));
  g_cus = prop.multiProcessorCount;
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return tb_set_error(TB_ERR_UNSUPPORTED, "device %d is %s; this library is built for gfx950 only", device,
                        prop.gcnArchName);
  return TB_OK;
}

int tb_device_cus(void) { return tb_num_cus(); }

// HIP-event timing of one kernel family on the stream it is launched on (bench.py roofline leg)
int tb_prof_enable(int kernel_id, int on) {
  if (kernel_id < 0 || kernel_id >= TB_K_COUNT) return tb_set_error(TB_ERR_ARG, "prof: bad kernel id");
  g_prof[kernel_id].on = on != 0;
  g_prof[kernel_id].n = 0;
  g_prof[kernel_id].open = false;
  g_prof[kernel_id].bytes = 0.0;
  return TB_OK;
}

// Call after a device synchronize.  Returns the number of timed launches; total_ms = sum of durations.
int tb_prof_read(int kernel_id, int* count, double* total_ms, double* algo_bytes) {
  if (kernel_id < 0 || kernel_id >= TB_K_COUNT) return tb_set_error(TB_ERR_ARG, "prof: bad kernel id");
  Prof& p = g_prof[kernel_id];
  double tot = 0.0;
  for (int i = 0; i < p.n; ++i) {
    float ms = 0.f;
    TB_HIP(hipEventElapsedTime(&ms, p.beg[i], p.end[i]));
    tot += ms;
  }
  if (count) *count = p.n;
  if (total_ms) *total_ms = tot;
  if (algo_bytes) *algo_bytes = p.bytes;
  p.n = 0;
  p.bytes = 0.0;
  return TB_OK;
}

// y[M,N] = act(x[M,K] . W[N,K]^T + b[N])      nn.Linear (+Tanh / +Sigmoid)
int tb_linear_act_fwd(const float* x, const float* W, const float* b, float* y, int M, int K, int N, int act,
                      void* stream) {
  if (!x || !W || !y) return tb_set_error(TB_ERR_ARG, "linear_act_fwd: null pointer");
  return tb_gemm_launch(x, K, 0, W, K, 0, y, N, b, M, N, K, act, 0, 1, nullptr, 0, static_cast<hipStream_t>(stream), 0);
}

size_t tb_colsum_workspace_bytes(int cols);
size_t tb_linear_workspace_bytes(int M, int K, int N) {
  (void)M; (void)K;
  return tb_gemm_ws_bytes() + tb_align_up(tb_colsum_workspace_bytes(N), 256) + 512;
}

// dz = dy * act'(y) must already have been applied by the caller when act != 0 (tb_act_bwd).
// dx[M,K] = dz.W ; dW[N,K] = dz^T.x ; db[N] = colsum(dz).  Any of dx/dW/db may be null.
int tb_colsum_acc(const float* a, long rows, int cols, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream);

int tb_linear_bwd(const float* dz, const float* x, const float* W, float* dx, float* dW, float* db, int M, int K,
                  int N, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!dz) return tb_set_error(TB_ERR_ARG, "linear_bwd: null pointer");
  if (ws_bytes < tb_linear_workspace_bytes(M, K, N)) return tb_set_error(TB_ERR_WORKSPACE, "linear_bwd: workspace too small");
  int rc;
  if (dx) {
    if (!W) return tb_set_error(TB_ERR_ARG, "linear_bwd: W is null");
    rc = tb_gemm_launch(dz, N, 0, W, K, 1, dx, K, nullptr, M, K, N, 0, 0, 1, nullptr, 0, s, 0);
    if (rc) return rc;
  }
  // dW's split-K slabs and db's column-sum partials are summed by ONE finish launch (4 launches for dW and db instead of 6)
  const size_t cs = tb_align_up(tb_colsum_workspace_bytes(N), 256);
  TbGemmDeferred d;
  d.splits = 0;
  if (dW) {
    if (!x) return tb_set_error(TB_ERR_ARG, "linear_bwd: x is null");
    rc = tb_gemm_launch_deferred(dz, N, 1, x, K, 1, dW, K, N, K, M, 0, 1,
                                 reinterpret_cast<float*>(static_cast<char*>(ws) + cs), ws_bytes - cs, s, accumulate, &d);
    if (rc) return rc;
  }
  int nparts = 0;
  if (db) {
    rc = tb_colsum_partials(dz, M, N, ws, cs, s, &nparts);
    if (rc) return rc;
  }
  return tb_linear_finish(&d, static_cast<const double*>(ws), nparts, N, db, accumulate, s);
}

}  // extern "C"
