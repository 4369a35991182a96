// K1: fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 fma chain).
//
// Replaces the library ops the reference reaches through torch: the LSTM input projection
// x.W_ih^T (aten::lstm, reference source/utils/lstm_layer.py:31-33,44-46,75,83), nn.Linear
// forward (lstm_layer.py:35-36,49-50,76,84-87) and the autograd matmuls behind them
// (dx = dY.W, dW = dY^T.X).
//
//   C[M,N] = act( opA(A)[M,K] . opB(B)[K,N] + bias[N] )
//
// LAY_A = 0: A stored [M][K] (k contiguous).   LAY_A = 1: A stored [K][M] (m contiguous).
// LAY_B = 0: B stored [N][K] (k contiguous).   LAY_B = 1: B stored [K][N] (n contiguous).
// Block tile 64x64x16, 4 waves as 2x2, each wave 2x2 MFMA tiles of 16x16.  Operand tiles are
// staged k-major in LDS with an 80-float row stride so the per-lane MFMA operand reads
// (lane -> [k = lane>>4][row = lane&15]) hit 32 distinct banks per half-wave.
// Split-K (gridDim.z) writes partial slabs that splitk_reduce sums in a fixed order
// (bitwise reproducible; no float atomics).
#include "common.h"

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LDT = 80;

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  int M, N, K;
  long lda, ldb, ldc;
  int act;
  int kchunk;
  int splits;
  float* slab;
  int b_shift, b_period;
  int vecA, vecB;
};

// One operand tile (64 rows x 16 k) global -> registers.
template <int LAY>
__device__ __forceinline__ void tile_load(const float* __restrict__ src, long ld, int r0, int rlim, int k0,
                                          int klim, int vec, int shift, int period, float (&v)[4]) {
  const int t = threadIdx.x;
  if (LAY == 0) {
    const int r = r0 + (t >> 2);
    const int k = k0 + (t & 3) * 4;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    if (r < rlim) {
      const float* p = src + (long)r * ld + k;
      if (vec && k + 3 < klim) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k + i < klim) v[i] = p[i];
      }
    }
  } else {
    int k = k0 + (t >> 4);
    const int r = r0 + (t & 15) * 4;
    v[0] = v[1] = v[2] = v[3] = 0.f;
    bool ok = k < klim;
    if (shift) {
      ok = ok && (k % period) >= shift;
      k -= shift;
    }
    if (ok) {
      const float* p = src + (long)k * ld + r;
      if (vec && r + 3 < rlim) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(p);
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (r + i < rlim) v[i] = p[i];
      }
    }
  }
}

template <int LAY>
__device__ __forceinline__ void tile_store(float* sm, const float (&v)[4]) {
  const int t = threadIdx.x;
  if (LAY == 0) {
    const int r = t >> 2, k = (t & 3) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[(k + i) * LDT + r] = v[i];
  } else {
    const int k = t >> 4, r = (t & 15) * 4;
    f32x4 x = {v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(&sm[k * LDT + r]) = x;
  }
}

template <int LAY_A, int LAY_B>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BK * LDT];
  __shared__ __attribute__((aligned(16))) float Bs[BK * LDT];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int kbeg = z * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  float ra[4], rb[4];
  if (kbeg < kend) {
    tile_load<LAY_A>(a.A, a.lda, m0, a.M, kbeg, kend, a.vecA, 0, 1, ra);
    tile_load<LAY_B>(a.B, a.ldb, n0, a.N, kbeg, kend, a.vecB, a.b_shift, a.b_period, rb);
  }
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    __syncthreads();  // previous tile fully consumed
    tile_store<LAY_A>(As, ra);
    tile_store<LAY_B>(Bs, rb);
    __syncthreads();
    if (k0 + BK < kend) {  // next tile's global loads fly under this tile's MFMAs
      tile_load<LAY_A>(a.A, a.lda, m0, a.M, k0 + BK, kend, a.vecA, 0, 1, ra);
      tile_load<LAY_B>(a.B, a.ldb, n0, a.N, k0 + BK, kend, a.vecB, a.b_shift, a.b_period, rb);
    }
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      const int kr = (ks * 4 + q) * LDT;
      const float a0 = As[kr + wm * 32 + col];
      const float a1 = As[kr + wm * 32 + 16 + col];
      const float b0 = Bs[kr + wn * 32 + col];
      const float b1 = Bs[kr + wn * 32 + 16 + col];
      acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  float* out = a.splits > 1 ? a.slab + (size_t)z * a.M * a.N : a.C;
  const long ldo = a.splits > 1 ? a.N : a.ldc;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = n0 + wn * 32 + ni * 16 + col;
      if (n >= a.N) continue;
      const float bv = (a.splits == 1 && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 32 + mi * 16 + q * 4 + r;
        if (m >= a.M) continue;
        float v = acc[mi][ni][r] + bv;
        if (a.splits == 1) {
          if (a.act == 1) v = tanhf(v);
          else if (a.act == 2) v = tb_sigmoid(v);
        }
        out[(long)m * ldo + n] = v;
      }
    }
}

// C[m][n] = act(sum_z slab[z][m][n] + bias[n]); fixed summation order.
__global__ void splitk_reduce_kernel(const float* __restrict__ slab, int splits, int M, int N, float* C,
                                     long ldc, const float* bias, int act) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slab[(size_t)z * total + i];
    const int m = (int)(i / N), n = (int)(i % N);
    if (bias) s += bias[n];
    if (act == 1) s = tanhf(s);
    else if (act == 2) s = tb_sigmoid(s);
    C[(long)m * ldc + n] = s;
  }
}

}  // namespace

// Host-side launcher shared by the C-ABI entry points (capi.hip, lstm.hip).
// transA: 0 -> A is [M][K]; 1 -> A is [K][M].   transB: 0 -> B is [N][K]; 1 -> B is [K][N].
int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream) {
  if (M <= 0 || N <= 0) return TB_OK;
  if (K <= 0) return tb_set_error(TB_ERR_ARG, "gemm: K must be positive");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.act = act;
  a.b_shift = b_shift;
  a.b_period = b_period > 0 ? b_period : 1;
  a.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  a.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  const long tiles = (long)mt * nt;
  int splits = 1;
  const int cus = tb_num_cus();
  if (tiles < cus && K >= 64 * BK) {
    splits = (int)((2L * cus + tiles - 1) / tiles);
    const int maxs = K / (16 * BK);
    if (splits > maxs) splits = maxs;
    if (splits > 64) splits = 64;
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    if (splits < 1) splits = 1;
  }
  int kchunk = (K + splits - 1) / splits;
  kchunk = (kchunk + BK - 1) / BK * BK;
  splits = (K + kchunk - 1) / kchunk;
  a.kchunk = kchunk;
  a.splits = splits;
  a.slab = ws;
  dim3 grid(mt, nt, splits);
  tb_prof_begin(TB_K_GEMM, stream);
  if (transA == 0 && transB == 0) hipLaunchKernelGGL((gemm_f32_kernel<0, 0>), grid, dim3(256), 0, stream, a);
  else if (transA == 0 && transB == 1) hipLaunchKernelGGL((gemm_f32_kernel<0, 1>), grid, dim3(256), 0, stream, a);
  else if (transA == 1 && transB == 1) hipLaunchKernelGGL((gemm_f32_kernel<1, 1>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<1, 0>), grid, dim3(256), 0, stream, a);
  TB_LAUNCH_CHECK("gemm_f32_kernel");
  if (splits > 1) {
    const long total = (long)M * N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, M, N, C, ldc, bias,
                       act);
    TB_LAUNCH_CHECK("splitk_reduce_kernel");
  }
  tb_prof_end(TB_K_GEMM, stream);
  return TB_OK;
}
