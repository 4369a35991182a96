// K6: single-output linear head  out[m] = act(x[m,:].w + b)  and its backward, for the discriminator's
// nn.Linear(hidden, 1) + Sigmoid (reference source/utils/lstm_layer.py:76,84-87) and any other N = 1 head.
//
// A GEMM tile is the wrong shape for one output column (a 256 x 32 tile with 31 dead columns, then a
// K = 1 outer product and a 1-row weight gradient through split-K slabs: ~280 us per pass at
// M = 131072).  This is a streaming job: forward reads x once (one wave per row, 16 bytes per lane, a
// wave reduction); backward reads x once and writes dx once, with dW accumulated in registers per
// thread (each thread owns 4 columns) and reduced over waves / blocks in a fixed order.  Plain fp32 FMA.
// Bound: HBM.  Algorithmic bytes: forward 4.M.K, backward 8.M.K.
#include "common.h"

namespace {

constexpr int kMaxChunks = 4;          // K <= 64 lanes * 4 floats * 4 chunks = 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

__device__ __forceinline__ float act_fwd(float z, int act) {
  if (act == 1) return tanhf(z);
  if (act == 2) return tb_sigmoid(z);
  return z;
}

template <int NC>
__global__ __launch_bounds__(256) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ b, float* __restrict__ out, long M,
                                                       int K, int act) {
  const int lane = threadIdx.x & 63;
  const long wave = blockIdx.x * 4L + (threadIdx.x >> 6), nwaves = gridDim.x * 4L;
  f32x4 wv[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = (c * 64 + lane) * 4;
    wv[c] = k < K ? *reinterpret_cast<const f32x4*>(w + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float bias = b ? b[0] : 0.f;
  // 8 rows per pass: 8 independent 1 KB loads in flight per wave, then ONE transposed butterfly for the eight row sums (the
  // xor-32 / 16 / 8 steps each hand half of the rows to the partner lane: 4 + 2 + 1 shuffles, then 3 for the last value,
  // instead of 8 x 6).  Every row's sum is added up along the same tree as a plain xor butterfly, so a row's result does
  // not depend on the slot it sits in.
  const bool b32 = lane & 32, b16 = lane & 16, b8 = lane & 8;
  for (long r0 = wave * 8; r0 < M; r0 += nwaves * 8) {
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      acc[i] = 0.f;
      const long r = r0 + i;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = (c * 64 + lane) * 4;
        if (r < M && k < K) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(x + r * K + k);
          acc[i] += xv[0] * wv[c][0] + xv[1] * wv[c][1] + xv[2] * wv[c][2] + xv[3] * wv[c][3];
        }
      }
    }
    float h4[4], h2[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) h4[k] = (b32 ? acc[4 + k] : acc[k]) + __shfl_xor(b32 ? acc[k] : acc[4 + k], 32, 64);
#pragma unroll
    for (int k = 0; k < 2; ++k) h2[k] = (b16 ? h4[2 + k] : h4[k]) + __shfl_xor(b16 ? h4[k] : h4[2 + k], 16, 64);
    float v = (b8 ? h2[1] : h2[0]) + __shfl_xor(b8 ? h2[0] : h2[1], 8, 64);
    v += __shfl_xor(v, 4, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 1, 64);
    const long r = r0 + (b32 ? 4 : 0) + (b16 ? 2 : 0) + (b8 ? 1 : 0);
    if ((lane & 7) == 0 && r < M) out[r] = act_fwd(v + bias, act);
  }
}

// dz[m] = dy[m] * act'(out[m]);  dx[m,:] = dz[m] * w;  partial dW / db per block
template <int NC>
__global__ __launch_bounds__(256) void head_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ out,
                                                       const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ dx, float* __restrict__ part, long M, int K,
                                                       int act, int need_dw) {
  __shared__ __attribute__((aligned(16))) float red[4][kMaxChunks * 256 + 4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const long wave = blockIdx.x * 4L + wid, nwaves = gridDim.x * 4L;
  f32x4 wv[NC], dwa[NC];
  float dba = 0.f;
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    const int k = (c * 64 + lane) * 4;
    wv[c] = (dx && k < K) ? *reinterpret_cast<const f32x4*>(w + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
    dwa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // 8 rows per pass, every load requested before the first use (with an early `break` per row the compiler issued one
  // row's loads at a time); rows are accumulated in row order, as before
  constexpr int RP = 8;
  for (long r0 = wave * RP; r0 < M; r0 += nwaves * RP) {
    float dz[RP];
    f32x4 xv[RP][NC];
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const long r = r0 + i;
      const bool ok = r < M;
      dz[i] = ok ? dy[r] : 0.f;
      const float y = (ok && act) ? out[r] : 0.f;
      if (act) dz[i] *= act == 1 ? (1.f - y * y) : y * (1.f - y);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = (c * 64 + lane) * 4;
        xv[i][c] = (need_dw && ok && k < K) ? *reinterpret_cast<const f32x4*>(x + r * K + k) : (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int i = 0; i < RP; ++i) {
      const long r = r0 + i;
      dba += dz[i];
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int k = (c * 64 + lane) * 4;
        dwa[c] += xv[i][c] * dz[i];
        if (dx && r < M && k < K) *reinterpret_cast<f32x4*>(dx + r * K + k) = wv[c] * dz[i];
      }
    }
  }
  // block partial: waves summed in order 0..3
#pragma unroll
  for (int c = 0; c < NC; ++c) *reinterpret_cast<f32x4*>(&red[wid][(c * 64 + lane) * 4]) = dwa[c];
  if (lane == 0) red[wid][kMaxChunks * 256] = dba;
  __syncthreads();
  float* p = part + (size_t)blockIdx.x * (K + 1);
  for (int k = threadIdx.x; k <= K; k += 256) {
    const int idx = k < K ? k : kMaxChunks * 256;
    p[k] = red[0][idx] + red[1][idx] + red[2][idx] + red[3][idx];
  }
}

// dw[k] / db = sum over the block partials, in a fixed order: one block per 4 columns, 64 threads per
// column each summing every 64th partial, then one thread per column adds the 64 sums in order.
__global__ __launch_bounds__(256) void head_reduce_kernel(const float* __restrict__ part, int nparts, int K, float* dw,
                                                          float* db, int accumulate) {
  __shared__ float sm[64][4];
  const int c = threadIdx.x & 3, g = threadIdx.x >> 2;
  const int k = blockIdx.x * 4 + c;
  float s = 0.f;
  if (k <= K)
    for (int p = g; p < nparts; p += 64) s += part[(size_t)p * (K + 1) + k];
  sm[g][c] = s;
  __syncthreads();
  if (threadIdx.x < 4 && k <= K) {
    float t = 0.f;
    for (int i = 0; i < 64; ++i) t += sm[i][c];
    if (k < K) {
      if (dw) dw[k] = accumulate ? dw[k] + t : t;
    } else if (db) {
      db[0] = accumulate ? db[0] + t : t;
    }
  }
}

int head_blocks_fwd(long M) {   // 8 rows per wave and pass
  long b = (M + 31) / 32;
  const long cap = 4L * tb_num_cus();
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

int head_blocks(long M) {
  long b = (M + 15) / 16;
  const long cap = 2L * tb_num_cus();
  return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

extern "C" {

int tb_head_supported(int K) { return (K > 0 && (K & 3) == 0 && K <= kMaxChunks * 256) ? 1 : 0; }

size_t tb_head_workspace_bytes(long M, int K) { return (size_t)head_blocks(M) * (K + 1) * sizeof(float) + 256; }

int tb_head_fwd(const float* x, const float* w, const float* b, float* out, long M, int K, int act, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M <= 0) return TB_OK;
  if (!x || !w || !out) return tb_set_error(TB_ERR_ARG, "head_fwd: null pointer");
  if (!tb_head_supported(K)) return tb_set_error(TB_ERR_UNSUPPORTED, "head_fwd: K=%d (needs K %% 4 == 0, K <= 1024)", K);
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w)) & 15)
    return tb_set_error(TB_ERR_ARG, "head_fwd: x and w must be 16-byte aligned");
  const int nc = (K + 255) / 256, grid = head_blocks_fwd(M);
  if (nc == 1) hipLaunchKernelGGL(head_fwd_kernel<1>, dim3(grid), dim3(256), 0, s, x, w, b, out, M, K, act);
  else if (nc == 2) hipLaunchKernelGGL(head_fwd_kernel<2>, dim3(grid), dim3(256), 0, s, x, w, b, out, M, K, act);
  else hipLaunchKernelGGL(head_fwd_kernel<4>, dim3(grid), dim3(256), 0, s, x, w, b, out, M, K, act);
  TB_LAUNCH_CHECK("head_fwd_kernel");
  return TB_OK;
}

// dy = gradient wrt out; out = forward result (read only when act != 0).  Any of dx / dw / db may be null.
int tb_head_bwd(const float* dy, const float* out, const float* x, const float* w, float* dx, float* dw, float* db,
                long M, int K, int act, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (M <= 0) return TB_OK;
  if (!dy || (act && !out) || (dw && !x) || (dx && !w) || !ws) return tb_set_error(TB_ERR_ARG, "head_bwd: null pointer");
  if (!tb_head_supported(K)) return tb_set_error(TB_ERR_UNSUPPORTED, "head_bwd: K=%d (needs K %% 4 == 0, K <= 1024)", K);
  if (ws_bytes < tb_head_workspace_bytes(M, K)) return tb_set_error(TB_ERR_WORKSPACE, "head_bwd: workspace too small");
  if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(dx)) & 15)
    return tb_set_error(TB_ERR_ARG, "head_bwd: x, w and dx must be 16-byte aligned");
  const int nc = (K + 255) / 256, grid = head_blocks(M);
  float* part = static_cast<float*>(ws);
  const int need_dw = dw ? 1 : 0;
  if (nc == 1) hipLaunchKernelGGL(head_bwd_kernel<1>, dim3(grid), dim3(256), 0, s, dy, out, x, w, dx, part, M, K, act, need_dw);
  else if (nc == 2) hipLaunchKernelGGL(head_bwd_kernel<2>, dim3(grid), dim3(256), 0, s, dy, out, x, w, dx, part, M, K, act, need_dw);
  else hipLaunchKernelGGL(head_bwd_kernel<4>, dim3(grid), dim3(256), 0, s, dy, out, x, w, dx, part, M, K, act, need_dw);
  TB_LAUNCH_CHECK("head_bwd_kernel");
  if (dw || db) {
    hipLaunchKernelGGL(head_reduce_kernel, dim3((K + 4) / 4), dim3(256), 0, s, part, grid, K, dw, db, accumulate);
    TB_LAUNCH_CHECK("head_reduce_kernel");
  }
  return TB_OK;
}

}  // extern "C"
