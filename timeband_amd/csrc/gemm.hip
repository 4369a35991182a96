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
//
// Block tile BM x BN x 16 with 4 waves laid out WGM x WGN, each wave a (BM/WGM) x (BN/WGN) patch of
// 16x16 MFMA tiles (128x128 as 2x2 waves of 4x4 tiles for wide outputs; 256x64 / 256x32 as 4x1 for
// the narrow ones), 32 flop per staged byte.  Operand tiles keep their global orientation in LDS
// (no transposing stores): a k-contiguous tile is [rows][16+4] and one ds_read_b128 per lane yields
// the operand of 4 k-steps, an m-contiguous tile is [16][rows+4] read with 4 ds_read_b32; both use
// the k order k = (lane>>4)*4 + step, so any layout pair multiplies correctly.  Two LDS stages, one
// barrier per k-tile, next tile's global loads in flight under the MFMAs.
// Split-K (gridDim.z) writes partial slabs that splitk_reduce sums in a fixed order (bitwise
// reproducible; no float atomics).
#include "common.h"

namespace {

constexpr int BK = 16;

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

template <int LAY, int ROWS>
struct Tile {
  static constexpr int LD = LAY == 0 ? BK + 4 : ROWS + 4;            // LDS row stride (floats)
  static constexpr int FLOATS = LAY == 0 ? ROWS * LD : BK * LD;      // one stage
  static constexpr int NCHUNK = ROWS * BK / 4;                       // 16-byte chunks of one stage
  static constexpr int NCH = (NCHUNK + 255) / 256;                   // chunks per thread
  static_assert(NCHUNK % 256 == 0 || NCHUNK < 256, "chunk count must tile the block");

  // chunk c of this thread -> (row, k) of its first element
  static __device__ __forceinline__ void where(int c, int& r, int& k) {
    const int e = threadIdx.x + 256 * c;
    if (LAY == 0) { r = e >> 2; k = (e & 3) * 4; }
    else { k = e / (ROWS / 4); r = (e % (ROWS / 4)) * 4; }
  }

  static __device__ __forceinline__ void load(const float* __restrict__ src, long ld, int r0, int rlim, int k0,
                                              int klim, int vec, int shift, int period, f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      r += r0; k += k0;
      v[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (NCHUNK < 256 && threadIdx.x >= NCHUNK) continue;
      if (LAY == 0) {
        if (r < rlim && k < klim) {
          const float* p = src + (long)r * ld + k;
          if (vec && k + 3 < klim) v[c] = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k + i < klim) v[c][i] = p[i];
          }
        }
      } else {
        bool ok = k < klim && r < rlim;
        if (shift) {   // B row k is source row k - shift, zero for the first `shift` rows of each period
          ok = ok && (k % period) >= shift;
          k -= shift;
        }
        if (ok) {
          const float* p = src + (long)k * ld + r;
          if (vec && r + 3 < rlim) v[c] = *reinterpret_cast<const f32x4*>(p);
          else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (r + i < rlim) v[c][i] = p[i];
          }
        }
      }
    }
  }

  static __device__ __forceinline__ void store(float* sm, const f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      if (NCHUNK < 256 && threadIdx.x >= NCHUNK) continue;
      if (LAY == 0) *reinterpret_cast<f32x4*>(sm + r * LD + k) = v[c];
      else *reinterpret_cast<f32x4*>(sm + k * LD + r) = v[c];
    }
  }

  // MFMA operand of the 16 rows starting at `row`, for the 4 k-steps of one k-tile
  static __device__ __forceinline__ f32x4 frag(const float* sm, int row, int col, int q) {
    if (LAY == 0) return *reinterpret_cast<const f32x4*>(sm + (row + col) * LD + q * 4);
    f32x4 f;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = sm[(q * 4 + i) * LD + row + col];
    return f;
  }
};

template <int LAY_A, int LAY_B, int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256, 2) void gemm_f32_kernel(GemmArgs a) {
  using TA = Tile<LAY_A, BM>;
  using TB = Tile<LAY_B, BN>;
  constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "bad wave layout");
  __shared__ __attribute__((aligned(16))) float As[2][TA::FLOATS];
  __shared__ __attribute__((aligned(16))) float Bs[2][TB::FLOATS];
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int kbeg = z * a.kchunk;
  const int kend = min(a.K, kbeg + a.kchunk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  f32x4 ra[TA::NCH], rb[TB::NCH];
  if (kbeg < kend) {
    TA::load(a.A, a.lda, m0, a.M, kbeg, kend, a.vecA, 0, 1, ra);
    TB::load(a.B, a.ldb, n0, a.N, kbeg, kend, a.vecB, a.b_shift, a.b_period, rb);
    TA::store(As[0], ra);
    TB::store(Bs[0], rb);
  }
  __syncthreads();
  int st = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {   // next tile's global loads fly under this tile's MFMAs
      TA::load(a.A, a.lda, m0, a.M, k0 + BK, kend, a.vecA, 0, 1, ra);
      TB::load(a.B, a.ldb, n0, a.N, k0 + BK, kend, a.vecB, a.b_shift, a.b_period, rb);
    }
    f32x4 af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = TA::frag(As[st], (wm * TM + i) * 16, col, q);
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = TB::frag(Bs[st], (wn * TN + j) * 16, col, q);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][kk], bf[j][kk], acc[i][j], 0, 0, 0);
    if (more) {
      TA::store(As[st ^ 1], ra);
      TB::store(Bs[st ^ 1], rb);
    }
    __syncthreads();
    st ^= 1;
  }

  float* out = a.splits > 1 ? a.slab + (size_t)z * a.M * a.N : a.C;
  const long ldo = a.splits > 1 ? a.N : a.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + (wn * TN + j) * 16 + col;
      if (n >= a.N) continue;
      const float bv = (a.splits == 1 && a.bias) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + (wm * TM + i) * 16 + q * 4 + r;
        if (m >= a.M) continue;
        float v = acc[i][j][r] + bv;
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

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
  if (transA == 0 && transB == 0) hipLaunchKernelGGL((gemm_f32_kernel<0, 0, BM, BN, WGM, WGN>), grid, dim3(256), 0, stream, a);
  else if (transA == 0 && transB == 1) hipLaunchKernelGGL((gemm_f32_kernel<0, 1, BM, BN, WGM, WGN>), grid, dim3(256), 0, stream, a);
  else if (transA == 1 && transB == 1) hipLaunchKernelGGL((gemm_f32_kernel<1, 1, BM, BN, WGM, WGN>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<1, 0, BM, BN, WGM, WGN>), grid, dim3(256), 0, stream, a);
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
  // tile shape by output width
  const int cfg = N <= 32 ? 0 : (N <= 64 ? 1 : 2);
  const int BM = cfg == 2 ? 128 : 256, BN = cfg == 0 ? 32 : (cfg == 1 ? 64 : 128);
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  const long tiles = (long)mt * nt;
  int splits = 1;
  const int cus = tb_num_cus();
  if (tiles < cus && K >= 64 * BK && ws) {
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
  if (cfg == 0) launch_cfg<256, 32, 4, 1>(a, transA, transB, grid, stream);
  else if (cfg == 1) launch_cfg<256, 64, 4, 1>(a, transA, transB, grid, stream);
  else launch_cfg<128, 128, 2, 2>(a, transA, transB, grid, stream);
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
