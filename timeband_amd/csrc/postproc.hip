// K5 window gather, K6 band (median / std over horizons), K7 adjust (imputation / anomaly scan),
// denormalise and the three scores -- the data side of the hot path.
//
// Reference: source/dataset.py:199-214 (windowing), :229-242 (denormalize), source/utils/dataset.py:17-23;
// source/trainer.py:238-287, source/cleaner.py:130-179, source/predictor.py:106-155 (adjust / predicts);
// source/metric.py:30-73.  Post-processing is float64 like the reference's numpy code; every
// multiply-add that numpy performs as two rounded operations is kept as two (no fma contraction).
#include "common.h"
#include <atomic>
#include <type_traits>

#pragma clang fp contract(off)

namespace {

// ---- K5 ----------------------------------------------------------------------------------
// window w: i = starts ? starts[w] : first + w*stride;
//   enc[w] = series[i-To : i, :]            (To*E contiguous floats)
//   dec[w] = series[i : i+Tf, target_cols]
// blockIdx.y strides the windows; the threads of a block copy one window's To*E encoder floats (16 bytes per
// thread when VEC: E % 4 == 0 and a 16-byte aligned series make every window start aligned) and gather its
// Tf*V decoder floats.  32-bit index arithmetic inside a window.
template <bool VEC>
__global__ __launch_bounds__(256) void window_gather_kernel(const float* __restrict__ series, int E,
                                                            const long* __restrict__ starts, long first, long stride,
                                                            int nwin, int To, int Tf, const int* __restrict__ tcols,
                                                            int V, float* __restrict__ enc, float* __restrict__ dec) {
  const int ne = To * E, nd = Tf * V;
  const int tid = blockIdx.x * blockDim.x + threadIdx.x, nthr = gridDim.x * blockDim.x;
  for (int w = blockIdx.y; w < nwin; w += gridDim.y) {
    const long i = starts ? starts[w] : first + (long)w * stride;
    const float* src = series + (i - To) * E;
    float* e = enc + (long)w * ne;
    if (VEC) {
      const f32x4* s4 = reinterpret_cast<const f32x4*>(src);
      f32x4* e4 = reinterpret_cast<f32x4*>(e);
      for (int r = tid; r < ne / 4; r += nthr) e4[r] = s4[r];
    } else {
      for (int r = tid; r < ne; r += nthr) e[r] = src[r];
    }
    const float* fut = series + i * E;
    float* d = dec + (long)w * nd;
    for (int k = tid; k < nd; k += nthr) {
      const int f = k / V, v = k - f * V;
      d[k] = fut[(long)f * E + tcols[v]];
    }
  }
}

// ---- denormalise (dataset.py:229-242) ----------------------------------------------------------
// fp32 0.5*(x+1); times fp64 (max-min); plus fp64 min; rounded back to fp32
__device__ __forceinline__ float denorm1(float x, double mn, double mx) {
  const float half = 0.5f * (x + 1.0f);
  const double t = (double)half * (mx - mn);
  return (float)(t + mn);
}

__global__ void denorm_kernel(const float* __restrict__ x, long n, int V, const double* __restrict__ dmin,
                              const double* __restrict__ dmax, float* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int v = (int)(i % V);
    const float half = 0.5f * (x[i] + 1.0f);
    const double delta = dmax[v] - dmin[v];
    const double t = (double)half * delta;
    out[i] = (float)(t + dmin[v]);
  }
}

// ---- min-max statistics and normalisation (dataset.py:297-321, :216-227) --------------------------------
// Column min / max over the rows of a row-major fp64 matrix.  Block (x, y): row slice x, columns [64y, 64y+64);
// the 256 threads are 4 row lanes x 64 columns, so a wave reads 64 consecutive doubles of a row.  NaNs are
// skipped (pandas' min/max).  Partials [slice][2][E], then one thread per column folds them in a fixed order.
__global__ __launch_bounds__(256) void minmax_partial_kernel(const double* __restrict__ data, long rows, int E,
                                                             double* __restrict__ part) {
  __shared__ double smn[256], smx[256];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + tx;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  double mn = __builtin_inf(), mx = -__builtin_inf();
  if (c < E)
    for (long r = r0 + ty; r < r1; r += 4) {
      const double v = data[r * E + c];
      if (v < mn) mn = v;     // false for NaN
      if (v > mx) mx = v;
    }
  smn[threadIdx.x] = mn;
  smx[threadIdx.x] = mx;
  __syncthreads();
  if (ty == 0 && c < E) {
    for (int k = 1; k < 4; ++k) {
      mn = fmin(mn, smn[k * 64 + tx]);
      mx = fmax(mx, smx[k * 64 + tx]);
    }
    part[((size_t)blockIdx.x * 2) * E + c] = mn;
    part[((size_t)blockIdx.x * 2 + 1) * E + c] = mx;
  }
}

__global__ void minmax_fold_kernel(const double* __restrict__ part, int nparts, int E, double* mn, double* mx, int init) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  double a = init ? __builtin_inf() : mn[c], b = init ? -__builtin_inf() : mx[c];
  for (int p = 0; p < nparts; ++p) {
    a = fmin(a, part[((size_t)p * 2) * E + c]);
    b = fmax(b, part[((size_t)p * 2 + 1) * E + c]);
  }
  mn[c] = a;
  mx[c] = b;
}

// the reference's two fix-ups, in its order: min == -inf -> max; then min == max -> max + 1 (dataset.py:311-312)
__global__ void minmax_finish_kernel(double* mn, double* mx, int E) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  double a = mn[c], b = mx[c];
  if (a == -__builtin_inf()) a = b;
  if (a == b) b = b + 1.0;
  mn[c] = a;
  mx[c] = b;
}

// 2 * ((x - min) / (max - min)) - 1 in fp64, numpy's operation order; fp32 series (what every window is cast to,
// utils/dataset.py:19-20) and, optionally, the fp64 frame (the reference's normalized.csv artefact)
__global__ void normalize_kernel(const double* __restrict__ data, long n, int E, const double* __restrict__ mn,
                                 const double* __restrict__ mx, float* __restrict__ out32, double* __restrict__ out64) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % E);
    const double q = (data[i] - mn[c]) / (mx[c] - mn[c]);
    const double v = 2.0 * q - 1.0;
    if (out32) out32[i] = (float)v;
    if (out64) out64[i] = v;
  }
}

// out[r][j] = data[r][cols[j]] : the target columns of the raw frame (dataset.py:130-133 observed / forecast)
__global__ void take_columns_kernel(const double* __restrict__ data, long rows, int E, const int* __restrict__ cols, int V,
                                    double* __restrict__ out) {
  const long n = rows * V;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / V;
    out[i] = data[r * E + cols[(int)(i - r * V)]];
  }
}

// ---- interpolate: ffill, bfill, 0 (dataset.py:111-115) on the device ------------------------------------
// A column is a chain along the rows, cut into blocks of FILL_ROWS rows: (A) every (block, column) thread finds the
// first and the last valid value of its block; (B) one thread per column walks the blocks: the value carried
// into each block (the column's first valid value for the leading gap -- the backward fill -- or 0 for an empty
// column); (C) every (block, column) thread fills its rows from the carried value and records which entries
// were missing.  Consecutive threads are consecutive columns, so every access is a contiguous run of a row.
constexpr int FILL_ROWS = 1024;

__global__ __launch_bounds__(64) void fill_scan_kernel(const double* __restrict__ data, long rows, int E,
                                                       double* __restrict__ first, double* __restrict__ last) {
  const int c = blockIdx.y * 64 + threadIdx.x;
  if (c >= E) return;
  const long r0 = (long)blockIdx.x * FILL_ROWS, r1 = r0 + FILL_ROWS < rows ? r0 + FILL_ROWS : rows;
  double f = __builtin_nan(""), l = __builtin_nan("");
  bool have = false;
  for (long r = r0; r < r1; ++r) {
    const double v = data[r * E + c];
    if (!isnan(v)) {
      if (!have) { f = v; have = true; }
      l = v;
    }
  }
  first[(size_t)blockIdx.x * E + c] = f;
  last[(size_t)blockIdx.x * E + c] = l;
}

__global__ void fill_carry_kernel(const double* __restrict__ first, double* __restrict__ last, int nblocks, int E) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= E) return;
  double carry = 0.0;                       // empty column -> 0
  for (int b = 0; b < nblocks; ++b) {
    const double f = first[(size_t)b * E + c];
    if (!isnan(f)) { carry = f; break; }    // leading gap -> the first valid value (backward fill)
  }
  for (int b = 0; b < nblocks; ++b) {       // last[b] becomes the value carried INTO block b
    const double l = last[(size_t)b * E + c];
    last[(size_t)b * E + c] = carry;
    if (!isnan(l)) carry = l;
  }
}

__global__ __launch_bounds__(64) void fill_apply_kernel(double* __restrict__ data, long rows, int E,
                                                        const double* __restrict__ carry_in, const int* __restrict__ tpos,
                                                        int V, long row_off, int* __restrict__ missing) {
  const int c = blockIdx.y * 64 + threadIdx.x;
  if (c >= E) return;
  const long r0 = (long)blockIdx.x * FILL_ROWS, r1 = r0 + FILL_ROWS < rows ? r0 + FILL_ROWS : rows;
  double prev = carry_in[(size_t)blockIdx.x * E + c];
  const int tp = (missing && tpos) ? tpos[c] : -1;
  for (long r = r0; r < r1; ++r) {
    const double v = data[r * E + c];
    const bool miss = isnan(v);
    if (miss) data[r * E + c] = prev;
    else prev = v;
    if (tp >= 0 && r >= row_off) missing[(r - row_off) * V + tp] = miss ? 1 : 0;
  }
}

// ---- K6 band -------------------------------------------------------------------------------
// One wave per (row r, variable v).  Row r (time offset from the batch's first forecast step)
// collects the predictions made for it at every horizon f by window w = r - f: from this batch
// (0 <= w < B) or from the history of the last Tf-1 windows (w < 0).  nanmedian / nanstd (ddof 0).
constexpr int BAND_WAVES = 4;

__global__ __launch_bounds__(64 * BAND_WAVES) void band_kernel(const float* __restrict__ pden, int B, int Tf, int V, int nrows,
                                                               const float* __restrict__ hist, int nhist,
                                                               const double* __restrict__ dmin, const double* __restrict__ dmax,
                                                               double* __restrict__ med, double* __restrict__ sd) {
  extern __shared__ float vals_all[];   // [BAND_WAVES][Tf]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * BAND_WAVES + wave;
  const long nitems = (long)nrows * V;
  float* vals = vals_all + (size_t)wave * Tf;
  __shared__ float s_med[BAND_WAVES][2];
  const bool live = item < nitems;
  const int r = live ? (int)(item / V) : 0, v = live ? (int)(item % V) : 0;
  // compacted gather (order inside `vals` is irrelevant for median and two-pass std); the count
  // is wave-uniform (every lane sees the same ballot), so it lives in a register
  int n = 0;
  if (live) {
    for (int f0 = 0; f0 < Tf; f0 += 64) {
      const int f = f0 + lane;
      float x = 0.f;
      bool ok = false;
      if (f < Tf) {
        const int w = r - f;
        if (w >= 0 && w < B) { x = pden[((long)w * Tf + f) * V + v]; ok = true; }
        else if (w < 0 && -w <= nhist) { x = hist[((long)(Tf - 1 + w) * Tf + f) * V + v]; ok = true; }
        if (ok && dmin) x = denorm1(x, dmin[v], dmax[v]);
        ok = ok && !isnan(x);
      }
      const unsigned long long m = __ballot(ok);
      if (ok) vals[n + __popcll(m & ((1ull << lane) - 1ull))] = x;
      n += __popcll(m);
    }
  }
  __syncthreads();
  if (live && n > 0) {
    // rank selection: element e has rank #{j : v_j < v_e or (v_j == v_e and j < e)}
    const int lo = (n - 1) >> 1, hi = n >> 1;
    for (int e = lane; e < n; e += 64) {
      const float x = vals[e];
      int rank = 0;
      for (int jj = 0; jj < n; ++jj) {
        const float y = vals[jj];
        rank += (y < x || (y == x && jj < e)) ? 1 : 0;
      }
      if (rank == lo) s_med[wave][0] = x;
      if (rank == hi) s_med[wave][1] = x;
    }
    double s = 0.0;
    for (int e = lane; e < n; e += 64) s += (double)vals[e];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const double mean = s / (double)n;
    double q = 0.0;
    for (int e = lane; e < n; e += 64) {
      const double d = (double)vals[e] - mean;
      q += d * d;
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
      const double a = (double)s_med[wave][0], b = (double)s_med[wave][1];
      med[item] = lo == hi ? a : (a + b) / 2.0;     // np.median: mean of the two middle values
      sd[item] = sqrt(q / (double)n);
    }
  } else if (live && lane == 0) {
    med[item] = __builtin_nan("");
    sd[item] = __builtin_nan("");
  }
}

// The same for forecast_len <= 64, one THREAD per (row, variable): consecutive lanes are consecutive
// variables of a row, so every gather load is a contiguous run of the wave (the wave-per-item kernel above
// reads 4 bytes per cache line), all loads of an item are in flight together, and the <= Tf values are
// sorted IN REGISTERS by a fully unrolled bitonic network (NP = Tf rounded up to a power of two; missing
// horizons and NaNs become +inf and sort to the end, so the first n sorted entries are the valid ones).
constexpr int BAND_T_MAX = 64;    // 128 values per thread no longer fit the register file

// Batcher's odd-even merge sort over P slots (a power of two) of which only the first NS exist: the missing ones stand
// for +inf, which a (low index <- min, high index <- max) comparator never moves, so every comparator that touches a
// slot >= NS is dropped (NS = 48: 384 comparators against 672 of a 64-wide bitonic network).  Fully unrolled: the
// values stay in registers.
template <int NS>
__device__ __forceinline__ void network_sort(float (&a)[NS]) {
  constexpr int P = NS <= 8 ? 8 : (NS <= 16 ? 16 : (NS <= 32 ? 32 : 64));
#pragma unroll
  for (int p = 1; p < P; p <<= 1) {
#pragma unroll
    for (int k = p; k >= 1; k >>= 1) {
#pragma unroll
      for (int j = k % p; j <= P - 1 - k; j += 2 * k) {
#pragma unroll
        for (int i = 0; i < k; ++i) {
          const int x = i + j, y = i + j + k;
          if (y < NS && x / (2 * p) == y / (2 * p)) {
            const float lo = fminf(a[x], a[y]), hi = fmaxf(a[x], a[y]);
            a[x] = lo;
            a[y] = hi;
          }
        }
      }
    }
  }
}

// median / population std of the n valid values among a[0..NS) (the others are +inf), and the band where it is final
template <int NS>
__device__ __forceinline__ void band_item_tail(float (&a)[NS], int n, long item, int r, double* __restrict__ med,
                                               double* __restrict__ sd, int direct_rows, double width,
                                               double* __restrict__ lower, double* __restrict__ upper) {
  double m_out, s_out;
  // Every slot of every lane valid (interior rows of a forecast_len that fills the slots: the bulk of a cleaning sweep):
  // the median's positions are constants and no sum needs a mask -- the same operations on the same values in the same
  // order as the general form below, without its ~100 compares, ~400 selects and the mask registers they pin.
  network_sort<NS>(a);   // (a lane without a valid value sorts its +inf: harmless, and ONE copy of the network in the code)
  if (__builtin_amdgcn_ballot_w64(n != NS) == 0) {
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) s += (double)a[i];
    const double mean = s / (double)NS;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const double d = (double)a[i] - mean;
      q += d * d;
    }
    constexpr int lo = (NS - 1) >> 1, hi = NS >> 1;
    m_out = lo == hi ? (double)a[lo] : ((double)a[lo] + (double)a[hi]) / 2.0;
    s_out = sqrt(q / (double)NS);
  } else if (n == 0) {
    m_out = __builtin_nan("");
    s_out = __builtin_nan("");
  } else {
    const int lo = (n - 1) >> 1, hi = n >> 1;
    float mlo = 0.f, mhi = 0.f;
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (i == lo) mlo = a[i];
      if (i == hi) mhi = a[i];
      s += i < n ? (double)a[i] : 0.0;
    }
    const double mean = s / (double)n;
    double q = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const double d = (double)a[i] - mean;
      q += i < n ? d * d : 0.0;
    }
    m_out = lo == hi ? (double)mlo : ((double)mlo + (double)mhi) / 2.0;   // np.median: mean of the two middle values
    s_out = sqrt(q / (double)n);
  }
  med[item] = m_out;
  sd[item] = s_out;
  if (r < direct_rows) {
    const double w = width * s_out;
    lower[item] = m_out - w;
    upper[item] = m_out + w;
  }
}

// Rows below `direct_rows` (those the std smear does not touch) get their band written here; band_finish_kernel does the rest.
template <int NS>
__global__ __launch_bounds__(256) void band_sort_kernel(const float* __restrict__ pden, int B, int Tf, int V, int nrows,
                                                        const float* __restrict__ hist, int nhist,
                                                        const double* __restrict__ dmin, const double* __restrict__ dmax,
                                                        double* __restrict__ med, double* __restrict__ sd,
                                                        int direct_rows, double width, double* __restrict__ lower,
                                                        double* __restrict__ upper) {
  const long item = (long)blockIdx.x * 256 + threadIdx.x;
  const long nitems = (long)nrows * V;
  if (item >= nitems) return;
  const int r = (int)(item / V), v = (int)(item % V);
  float a[NS];
  int n = 0;
  const double mn = dmin ? dmin[v] : 0.0, mx = dmin ? dmax[v] : 0.0;
#pragma unroll
  for (int f = 0; f < NS; ++f) {
    // branch-free: every lane loads (from a clamped address where the horizon does not exist) and the value is picked
    // afterwards -- with per-lane branches around the loads hipcc waits for each load before it issues the next one, and
    // the forecast_len dependent gathers of an item ran one HBM latency after the other
    const int w = r - f;
    const bool inb = f < Tf && w >= 0 && w < B;
    const bool inh = f < Tf && w < 0 && -w <= nhist && hist != nullptr;
    const float* src = inh ? hist : pden;
    const long off = inb ? ((long)w * Tf + f) * V + v : (inh ? ((long)(Tf - 1 + w) * Tf + f) * V + v : (long)v);
    float x = (B > 0 || inh) ? src[off] : 0.f;
    x = (inb || inh) ? x : __builtin_nanf("");
    if (dmin) x = denorm1(x, mn, mx);   // NaN stays NaN
    const bool ok = !isnan(x);
    n += ok ? 1 : 0;
    a[f] = ok ? x : __builtin_inff();
  }
  band_item_tail<NS>(a, n, item, r, med, sd, direct_rows, width, lower, upper);
}

// ---- K6b last linear layer + band in one kernel (cleaning pass, forecast_len <= 48, hidden <= 64) ----------------------
// The generator ends in out[w][f][:] = act(hid[w][f][:] . W^T + b) (lstm_layer.py:50: Linear(hidden, out_dim) + Tanh over the
// (windows x forecast_len) rows), and the band of series row r is taken over out[r - f][f][:], f = 0 .. Tf-1.  Written to
// HBM and read back, `out` is the largest tensor of the pass (805 MB per 8192 windows at 512 variables) and nothing else
// reads it but the band history (the last Tf-1 windows, kept by the caller).  Here a workgroup takes a row r, gathers its
// Tf hidden rows hid[r - f][f][0..KD) -- each hidden row belongs to exactly one r --, forms the Tf x V forecasts on the
// matrix cores and sorts them in registers.  Wave w of the 8 owns variables [64 w, 64 w + 64) (+ 512 blockIdx.y): its
// W rows stay in registers as split bf16 for the whole launch.  The products replicate the GEMM kernel's arithmetic for
// this shape operation by operation (first operand W, k-tiles of 32 ascending, lo.hi, hi.lo, hi.hi, then bias, then tb_tanh),
// so the forecasts -- and with them the band -- are the same bits as on the two-kernel path.
template <int NS, int KD>
__global__ __launch_bounds__(512, 1) void dimband_kernel(const float* __restrict__ hid, int B, int Tf, const float* __restrict__ W,
                                                         const float* __restrict__ bias, int act, int V, int nrows,
                                                         const float* __restrict__ hist, int nhist,
                                                         const double* __restrict__ dmin, const double* __restrict__ dmax,
                                                         double* __restrict__ med, double* __restrict__ sd, int direct_rows,
                                                         double width, double* __restrict__ lower, double* __restrict__ upper) {
  constexpr int NFT = (NS + 15) / 16;          // 16-wide forecast tiles
  constexpr int FR = NFT * 16;                 // rows of the hidden tile
  constexpr int NKS = KD / 32;                 // 32-deep k-steps
  constexpr int LDA = KD + 8;                  // hidden tile row stride (bf16 elements)
  constexpr int PLD = 64 + 4;                  // forecast patch row stride (floats)
  constexpr int NCHUNK = FR * KD / 4, NCH = (NCHUNK + 511) / 512;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds_db[];
  // [2 buffers][hi | lo][FR][LDA], then per wave a [FR][PLD] patch of forecasts
  float* patch_all = reinterpret_cast<float*>(lds_db + 2 * 2 * FR * LDA);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, q = lane >> 4;
  const int vbase = blockIdx.y * 512 + wave * 64;
  float* patch = patch_all + wave * FR * PLD;
  const __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc((void*)hid, 0, (int)((long)B * Tf * KD * 4), 0x00020000);
  // W rows of this wave's 64 variables as the MFMA's first operand: lane (col, q) holds W[v = vt*16 + col][k = s*32 + q*8 ..]
  bf16x8 wh[4][NKS], wl[4][NKS];
  f32x4 bv[4];
#pragma unroll
  for (int vt = 0; vt < 4; ++vt) {
    const int v = vbase + vt * 16 + col;
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      f32x4 w0 = {0.f, 0.f, 0.f, 0.f}, w1 = {0.f, 0.f, 0.f, 0.f};
      if (v < V) {
        w0 = *reinterpret_cast<const f32x4*>(W + (long)v * KD + s * 32 + q * 8);
        w1 = *reinterpret_cast<const f32x4*>(W + (long)v * KD + s * 32 + q * 8 + 4);
      }
      tb_split8(w0, w1, wh[vt][s], wl[vt][s]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vb = vbase + vt * 16 + q * 4 + i;
      bv[vt][i] = (bias && vb < V) ? bias[vb] : 0.f;
    }
  }
  const int v = vbase + lane;                  // the variable this lane sorts
  const bool vok = v < V;
  const double mn = (dmin && vok) ? dmin[v] : 0.0, mx = (dmin && vok) ? dmax[v] : 0.0;

  auto load_rows = [&](int r, f32x4 (&ra)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = tid + 512 * c, f = e / (KD / 4), k4 = e % (KD / 4);
      const int w = r - f;
      const bool ok = e < NCHUNK && f < Tf && w >= 0 && w < B && r < nrows;
      ra[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rH, ok ? (int)((((long)w * Tf + f) * KD + k4 * 4) * 4) : 0x7FFFFFF0, 0, 0));
    }
  };
  auto store_rows = [&](int b, const f32x4 (&ra)[NCH]) {
    unsigned short* hi = lds_db + b * 2 * FR * LDA;
    unsigned short* lo = hi + FR * LDA;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int e = tid + 512 * c, f = e / (KD / 4), k4 = e % (KD / 4);
      if (e < NCHUNK) {
        u32x2 sh, sl;
        tb_split4(ra[c], sh, sl);
        *reinterpret_cast<u32x2*>(hi + f * LDA + k4 * 4) = sh;
        *reinterpret_cast<u32x2*>(lo + f * LDA + k4 * 4) = sl;
      }
    }
  };
  f32x4 ra[NCH];
  int r = blockIdx.x;
  load_rows(r, ra);
  store_rows(0, ra);
  __syncthreads();
  int b = 0;
  for (; r < nrows; r += gridDim.x, b ^= 1) {
    load_rows(r + gridDim.x, ra);              // the next row's hidden rows, in flight under this row's work
    const unsigned short* Ah = lds_db + b * 2 * FR * LDA;
    const unsigned short* Al = Ah + FR * LDA;
    f32x4 acc[4][NFT];
#pragma unroll
    for (int vt = 0; vt < 4; ++vt)
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) acc[vt][ft] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
#pragma unroll
      for (int ft = 0; ft < NFT; ++ft) {
        const int ao = (ft * 16 + col) * LDA + s * 32 + q * 8;
        const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Ah + ao));
        const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(Al + ao));
#pragma unroll
        for (int vt = 0; vt < 4; ++vt) acc[vt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[vt][s], ah, acc[vt][ft], 0, 0, 0);
#pragma unroll
        for (int vt = 0; vt < 4; ++vt) acc[vt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[vt][s], al, acc[vt][ft], 0, 0, 0);
#pragma unroll
        for (int vt = 0; vt < 4; ++vt) acc[vt][ft] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[vt][s], ah, acc[vt][ft], 0, 0, 0);
      }
    }
    // bias, activation; lane (col, q) holds variables vt*16 + q*4 .. +3 of forecast step ft*16 + col: into the wave's patch
    // (ONE branch on the activation around the whole block: inside the loops it is four scalar branches per value)
    auto emit = [&](auto kind) {
#pragma unroll
      for (int vt = 0; vt < 4; ++vt)
#pragma unroll
        for (int ft = 0; ft < NFT; ++ft) {
          f32x4 o = acc[vt][ft];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            o[i] += bv[vt][i];
            if (decltype(kind)::value == 1) o[i] = tb_tanh(o[i]);
            else if (decltype(kind)::value == 2) o[i] = tb_sigmoid(o[i]);
          }
          *reinterpret_cast<f32x4*>(patch + (ft * 16 + col) * PLD + vt * 16 + q * 4) = o;
        }
    };
    if (act == 1) emit(std::integral_constant<int, 1>{});
    else if (act == 2) emit(std::integral_constant<int, 2>{});
    else emit(std::integral_constant<int, 0>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // this lane's variable: its forecasts for row r at every horizon (window r - f; the history / NaN where that
    // window is not in this batch), denormalised like the two-kernel path
    float a[NS];
#pragma unroll
    for (int f = 0; f < NS; ++f) a[f] = patch[f * PLD + lane];      // all reads in flight together
    if (r < Tf - 1 || r >= B || Tf < NS) {     // some horizon of this row has no window in the batch (workgroup-uniform)
#pragma unroll
      for (int f = 0; f < NS; ++f) {
        const int w = r - f;
        const bool inb = f < Tf && w >= 0 && w < B;
        const bool inh = f < Tf && w < 0 && -w <= nhist && hist != nullptr && vok;
        const float hx = inh ? hist[((long)(Tf - 1 + w) * Tf + f) * V + v] : 0.f;
        a[f] = inb ? a[f] : (inh ? hx : __builtin_nanf(""));
      }
    }
    int n = 0;
#pragma unroll
    for (int f = 0; f < NS; ++f) {
      float x = a[f];
      if (dmin) x = denorm1(x, mn, mx);   // NaN stays NaN
      const bool ok = !isnan(x);
      n += ok ? 1 : 0;
      a[f] = ok ? x : __builtin_inff();
    }
    __builtin_amdgcn_wave_barrier();           // the patch is free for the next row
    if (vok) band_item_tail<NS>(a, n, (long)r * V + v, r, med, sd, direct_rows, width, lower, upper);
    store_rows(b ^ 1, ra);
    __syncthreads();
  }
}

// std smear over the last Tf rows (trainer.py:280-282): a chain along the rows, one thread per variable
__global__ void band_smear_kernel(double* __restrict__ sd, int R, int Tf, int V) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  for (int f = Tf - 1; f >= 1; --f) {
    const double gamma = (double)(Tf - f) / (double)(Tf - 1);
    const double add = sd[(long)(R - f - 1) * V + v] * gamma;
    sd[(long)(R - f) * V + v] = sd[(long)(R - f) * V + v] + add;
  }
}

// ... then the band, elementwise
__global__ void band_finish_kernel(const double* __restrict__ med, const double* __restrict__ sd, long first, long n, double width,
                                   double* __restrict__ lower, double* __restrict__ upper) {
  for (long i = first + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double m = med[i], w = width * sd[i];
    lower[i] = m - w;
    upper[i] = m + w;
  }
}

// history = the last Tf-1 windows of (old history ++ this batch)
__global__ void hist_update_kernel(const float* __restrict__ pden, int B, int Tf, int V,
                                   const float* __restrict__ hist_in, float* __restrict__ hist_out) {
  const long per = (long)Tf * V;
  const long total = (long)(Tf - 1) * per;
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const int k = (int)(g / per);
    const long r = g % per;
    const int src = k + B;   // index in (hist ++ pden)
    hist_out[g] = src < Tf - 1 ? hist_in[(long)src * per + r] : pden[(long)(src - (Tf - 1)) * per + r];
  }
}

// ---- K7 adjust ---------------------------------------------------------------------------------
// variant 0: trainer.py:238-257, 1: cleaner.py:130-149, 2: predictor.py:106-125.
// One row of the scan: the adjusted value of row p from its inputs and the adjusted value of row p-1.
__device__ __forceinline__ double adjust_row(int variant, double value, double m, double pr, double lo, double up,
                                             double prev, double a, double b) {
  const double fill = a * pr + (1.0 - a) * prev;
  if (variant == 0) {
    value = (1.0 - m) * value + m * fill;
    const double lm = (value < lo ? 1.0 : 0.0) * (1.0 - m);
    const double um = (value > up ? 1.0 : 0.0) * (1.0 - m);
    value = (1.0 - lm) * value + lm * (b * pr + (1.0 - b) * value);
    value = (1.0 - um) * value + um * (b * pr + (1.0 - b) * value);
  } else if (variant == 1) {
    const double lm = value < lo ? 1.0 : 0.0, um = value > up ? 1.0 : 0.0;
    value = (1.0 - m) * value + m * fill;
    value = (1.0 - lm) * value + lm * (b * pr + (1.0 - b) * value);
    value = (1.0 - um) * value + um * (b * pr + (1.0 - b) * value);
  } else {
    const double lm = value < lo ? 1.0 : 0.0, um = value > up ? 1.0 : 0.0;
    value = (1.0 - lm) * value + lm * (b * pr + (1.0 - b) * value);
    value = (1.0 - um) * value + um * (b * pr + (1.0 - b) * value);
    value = (1.0 - m) * value + m * fill;
  }
  return value;
}

constexpr int ADJ_ROWS = 16;    // rows whose inputs are fetched together (the chain waits for memory once per batch)
constexpr int ADJ_CHUNK = 64;   // rows per thread of the parallel form
constexpr int ADJ_FLAGS = 64;
__device__ int tb_adjust_flags[ADJ_FLAGS];   // per call: 1 if an input is not finite (host picks the slot round-robin)

__global__ void adjust_flag_reset_kernel(int slot) { tb_adjust_flags[slot] = 0; }

__global__ void adjust_flag_kernel(int slot, const double* __restrict__ seed, const double* __restrict__ reals,
                                   const double* __restrict__ preds, const double* __restrict__ lower,
                                   const double* __restrict__ upper, long n, int V) {
  bool bad = false;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    bad = bad || !isfinite(reals[i]) || !isfinite(preds[i]) || !isfinite(lower[i]) || !isfinite(upper[i]);
    if (i < V) bad = bad || !isfinite(seed[i]);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(&tb_adjust_flags[slot], 1);
}

// Sequential in p (each row uses the already-adjusted previous row), parallel over variables: the literal
// order of the reference.  Runs only when an input is not finite (a NaN then travels down the column through
// every row, masked or not, exactly as in the reference's arithmetic).
__global__ __launch_bounds__(64) void adjust_kernel(int slot, int variant, const double* __restrict__ seed,
                                                    const double* __restrict__ reals, const double* __restrict__ preds,
                                                    const int* __restrict__ masks, const double* __restrict__ lower,
                                                    const double* __restrict__ upper, int n, int V, double a, double b,
                                                    double* __restrict__ target) {
  if (slot >= 0 && tb_adjust_flags[slot] == 0) return;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double prev = seed[v];
  for (int p0 = 0; p0 < n; p0 += ADJ_ROWS) {
    double rv_[ADJ_ROWS], pr_[ADJ_ROWS], lo_[ADJ_ROWS], up_[ADJ_ROWS];
    int mk_[ADJ_ROWS];
#pragma unroll
    for (int i = 0; i < ADJ_ROWS; ++i) {
      const long o = (long)(p0 + i < n ? p0 + i : n - 1) * V + v;
      rv_[i] = reals[o]; pr_[i] = preds[o]; lo_[i] = lower[o]; up_[i] = upper[o]; mk_[i] = masks[o];
    }
#pragma unroll
    for (int i = 0; i < ADJ_ROWS; ++i) {
      const int p = p0 + i;
      if (p >= n) continue;
      prev = adjust_row(variant, rv_[i], (double)mk_[i], pr_[i], lo_[i], up_[i], prev, a, b);
      target[(long)p * V + v] = prev;
    }
  }
}

// The same scan, parallel over chunks of rows.  A row's value depends on the previous row's only through
// m * fill: with finite inputs a row whose mask is 0 does not depend on it at all (0 * fill adds nothing).
// So the thread of (chunk, variable) walks back from its first row to the nearest row with mask 0 (or to the
// seed), and replays the reference's arithmetic from there, in the reference's order: the same operations
// on the same operands as the sequential form, hence the same values (up to the sign of a zero).
__global__ __launch_bounds__(64) void adjust_chunk_kernel(int slot, int variant, const double* __restrict__ seed,
                                                          const double* __restrict__ reals,
                                                          const double* __restrict__ preds, const int* __restrict__ masks,
                                                          const double* __restrict__ lower,
                                                          const double* __restrict__ upper, int n, int V, double a,
                                                          double b, double* __restrict__ target) {
  if (tb_adjust_flags[slot] != 0) return;
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  const int p0 = blockIdx.y * ADJ_CHUNK;
  int q = p0 - 1;
  while (q >= 0 && masks[(long)q * V + v] != 0) --q;
  double prev = q < 0 ? seed[v] : 0.0;   // row q (mask 0) does not look at it
  for (int p = q < 0 ? 0 : q; p < p0; ++p) {
    const long o = (long)p * V + v;
    prev = adjust_row(variant, reals[o], (double)masks[o], preds[o], lower[o], upper[o], prev, a, b);
  }
  const int pend = p0 + ADJ_CHUNK < n ? p0 + ADJ_CHUNK : n;
  for (int pb = p0; pb < pend; pb += ADJ_ROWS) {
    double rv_[ADJ_ROWS], pr_[ADJ_ROWS], lo_[ADJ_ROWS], up_[ADJ_ROWS];
    int mk_[ADJ_ROWS];
#pragma unroll
    for (int i = 0; i < ADJ_ROWS; ++i) {
      const long o = (long)(pb + i < pend ? pb + i : pend - 1) * V + v;
      rv_[i] = reals[o]; pr_[i] = preds[o]; lo_[i] = lower[o]; up_[i] = upper[o]; mk_[i] = masks[o];
    }
#pragma unroll
    for (int i = 0; i < ADJ_ROWS; ++i) {
      const int p = pb + i;
      if (p >= pend) continue;
      prev = adjust_row(variant, rv_[i], (double)mk_[i], pr_[i], lo_[i], up_[i], prev, a, b);
      target[(long)p * V + v] = prev;
    }
  }
}

// ---- scores (metric.py:30-73) ------------------------------------------------------------------
// out[0..2] = nme, nmae, rmse of this batch; running[0..2] += them
__global__ void metric_kernel(const double* __restrict__ tru, const double* __restrict__ pred,
                              const int* __restrict__ mask, long n, double esp, double* out, double* running) {
  __shared__ double sh[5][4];
  double s[5] = {0, 0, 0, 0, 0};   // sq err, count unmasked, sum ne, sum |ne|, count nonzero
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    if (mask[i] != 0) continue;
    const double t = tru[i], p = pred[i];
    s[0] += (t - p) * (t - p);
    s[1] += 1.0;
    if (t != 0.0) {
      const double ne = (t - p) / (esp + t);
      s[2] += ne;
      s[3] += fabs(ne);
      s[4] += 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 5; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) sh[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double t[5];
    for (int k = 0; k < 5; ++k) t[k] = sh[k][0] + sh[k][1] + sh[k][2] + sh[k][3];
    const double nme = t[2] / t[4], nmae = t[3] / t[4], rmse = sqrt(t[0] / t[1]);
    out[0] = nme; out[1] = nmae; out[2] = rmse;
    if (running) { running[0] += nme; running[1] += nmae; running[2] += rmse; }
  }
}

// The same in two stages for data-parallel runs: the five sums of a batch's rows are additive over the ranks
// (each holds the rows of its windows), the three scores are not.  sums = [sq err, unmasked, sum ne, sum |ne|, nonzero].
__global__ void metric_sums_kernel(const double* __restrict__ tru, const double* __restrict__ pred,
                                   const int* __restrict__ mask, long n, double esp, double* sums) {
  __shared__ double sh[5][4];
  double s[5] = {0, 0, 0, 0, 0};
  for (long i = threadIdx.x; i < n; i += blockDim.x) {
    if (mask[i] != 0) continue;
    const double t = tru[i], p = pred[i];
    s[0] += (t - p) * (t - p);
    s[1] += 1.0;
    if (t != 0.0) {
      const double ne = (t - p) / (esp + t);
      s[2] += ne;
      s[3] += fabs(ne);
      s[4] += 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = 0; k < 5; ++k) {
    double v = s[k];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if (lane == 0) sh[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < 5) sums[threadIdx.x] = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
}

// scores of k batches from their (globally summed) sums, in batch order: out[b][0..2], running[0..2] += them
__global__ void metric_finish_kernel(const double* __restrict__ sums, int k, double* out, double* running) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int b = 0; b < k; ++b) {
    const double* t = sums + 5 * b;
    const double nme = t[2] / t[4], nmae = t[3] / t[4], rmse = sqrt(t[0] / t[1]);
    if (out) { out[3 * b] = nme; out[3 * b + 1] = nmae; out[3 * b + 2] = rmse; }
    if (running) { running[0] += nme; running[1] += nmae; running[2] += rmse; }
  }
}

inline int grid_for(long n, int cap = 4096) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" {

int tb_window_gather(const float* series, long series_len, int E, const long* starts, long first, long stride,
                     int nwin, int To, int Tf, const int* target_cols, int V, float* enc, float* dec,
                     void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (nwin <= 0) return TB_OK;
  if (!series || !target_cols || !enc || !dec) return tb_set_error(TB_ERR_ARG, "window_gather: null pointer");
  if (!starts) {   // host-checkable bounds: window w reads rows [i-To, i+Tf)
    const long last = first + (long)(nwin - 1) * stride;
    if (first - To < 0 || last + Tf > series_len || stride < 0)
      return tb_set_error(TB_ERR_ARG, "window_gather: windows [%ld..%ld] leave the series (len %ld)", first, last,
                          series_len);
  }
  if ((long)To * E > 0x7FFFFFFFl || (long)Tf * V > 0x7FFFFFFFl) return tb_set_error(TB_ERR_UNSUPPORTED, "window_gather: window too large");
  const long per = (long)To * E > (long)Tf * V ? (long)To * E : (long)Tf * V;
  const bool vec = (E & 3) == 0 && (reinterpret_cast<uintptr_t>(series) & 15) == 0 && (reinterpret_cast<uintptr_t>(enc) & 15) == 0;
  int bx = (int)(((vec ? per / 4 : per) + 255) / 256);
  if (bx > 16) bx = 16;
  if (bx < 1) bx = 1;
  const dim3 grid(bx, nwin < 32768 ? nwin : 32768);
  tb_prof_begin(TB_K_GATHER, s);
  if (vec) hipLaunchKernelGGL(window_gather_kernel<true>, grid, dim3(256), 0, s, series, E, starts, first, stride, nwin, To,
                              Tf, target_cols, V, enc, dec);
  else hipLaunchKernelGGL(window_gather_kernel<false>, grid, dim3(256), 0, s, series, E, starts, first, stride, nwin, To,
                          Tf, target_cols, V, enc, dec);
  TB_LAUNCH_CHECK("window_gather_kernel");
  tb_prof_end(TB_K_GATHER, s);
  return TB_OK;
}

int tb_denormalize(const float* x, long n, int V, const double* dmin, const double* dmax, float* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0) return TB_OK;
  if (!x || !dmin || !dmax || !out) return tb_set_error(TB_ERR_ARG, "denormalize: null pointer");
  hipLaunchKernelGGL(denorm_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, n, V, dmin, dmax, out);
  TB_LAUNCH_CHECK("denorm_kernel");
  return TB_OK;
}

}  // extern "C"
namespace {
// the std smear over the last Tf rows and the band of the rows the sort kernel left (those the smear changes)
int band_smear_finish(double* sd, const double* med, double* lower, double* upper, int R, int Tf, int V, int rows_out, int smear,
                      int direct_rows, double band_width, hipStream_t s) {
  const long nitems = (long)rows_out * V;
  if (smear) {
    hipLaunchKernelGGL(band_smear_kernel, dim3((V + 63) / 64), dim3(64), 0, s, sd, R, Tf, V);
    TB_LAUNCH_CHECK("band_smear_kernel");
  }
  if (direct_rows < rows_out) {
    const long first = (long)direct_rows * V;
    hipLaunchKernelGGL(band_finish_kernel, dim3(grid_for(nitems - first)), dim3(256), 0, s, med, sd, first, nitems, band_width,
                       lower, upper);
    TB_LAUNCH_CHECK("band_finish_kernel");
  }
  return TB_OK;
}

template <int NS, int KD>
int launch_dimband(dim3 g, size_t lds, hipStream_t s, const float* hid, int B, int Tf, const float* W, const float* bias, int act,
                   int V, int rows_out, const float* hist_in, int nhist, const double* dmin, const double* dmax, double* med,
                   double* sd, int direct_rows, double width, double* lower, double* upper) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&dimband_kernel<NS, KD>), hipFuncAttributeMaxDynamicSharedMemorySize,
                               160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((dimband_kernel<NS, KD>), g, dim3(512), lds, s, hid, B, Tf, W, bias, act, V, rows_out, hist_in, nhist, dmin,
                     dmax, med, sd, direct_rows, width, lower, upper);
  TB_LAUNCH_CHECK("dimband_kernel");
  return TB_OK;
}
}  // namespace
extern "C" {

// pden: (B,Tf,V) predictions of this batch.  hist_in/hist_out: (Tf-1,Tf,V) predictions of the last Tf-1
// windows before / after this batch (distinct buffers), nhist = valid windows in hist_in.
// med/sd/lower/upper: float64 outputs of `rows_out` rows (sd is returned AFTER the smear).
// tb_band_update_part: only the first rows_out <= B+Tf-1 rows, the std smear over the last Tf rows optional
// (smear needs rows_out == B+Tf-1).  A data-parallel rank that is not the last of its global batch computes
// its own B rows without the smear (the rows after them belong to its right neighbour).
int tb_band_update_part(const float* pden, int B, int Tf, int V, const float* hist_in, int nhist, float* hist_out,
                        const double* dmin, const double* dmax, double band_width, double* med, double* sd,
                        double* lower, double* upper, int rows_out, int smear, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0) return TB_OK;
  if (Tf < 2) return tb_set_error(TB_ERR_ARG, "band_update: forecast_len must be >= 2 (the reference divides by Tf-1)");
  if (!pden || !hist_in || !med || !sd || !lower || !upper || hist_in == hist_out)
    return tb_set_error(TB_ERR_ARG, "band_update: bad pointer");
  if (nhist < 0 || nhist > Tf - 1) return tb_set_error(TB_ERR_ARG, "band_update: nhist out of range");
  if ((dmin == nullptr) != (dmax == nullptr)) return tb_set_error(TB_ERR_ARG, "band_update: dmin and dmax go together");
  const int R = B + Tf - 1;
  if (rows_out < 1 || rows_out > R) return tb_set_error(TB_ERR_ARG, "band_update: rows_out %d outside [1, %d]", rows_out, R);
  if (smear && rows_out != R) return tb_set_error(TB_ERR_ARG, "band_update: the std smear needs every row (rows_out == B+Tf-1)");
  const long nitems = (long)rows_out * V;
  const size_t lds = (size_t)BAND_WAVES * Tf * sizeof(float);
  if (lds > 60 * 1024) return tb_set_error(TB_ERR_UNSUPPORTED, "band_update: forecast_len %d too long", Tf);
  tb_prof_begin(TB_K_BAND, s);
  int direct_rows = 0;   // rows whose band the sort kernel writes itself (the smear changes the std of the last Tf rows only)
  if (Tf <= BAND_T_MAX) {
    const dim3 g((unsigned)((nitems + 255) / 256));
    direct_rows = smear ? (R - Tf > 0 ? R - Tf : 0) : rows_out;
#define TB_BAND_CASE(ns)                                                                                                  \
  if (Tf <= ns) {                                                                                                          \
    hipLaunchKernelGGL(band_sort_kernel<ns>, g, dim3(256), 0, s, pden, B, Tf, V, rows_out, hist_in, nhist, dmin, dmax, med, \
                       sd, direct_rows, band_width, lower, upper);                                                          \
  } else
    TB_BAND_CASE(8) TB_BAND_CASE(16) TB_BAND_CASE(24) TB_BAND_CASE(32) TB_BAND_CASE(40) TB_BAND_CASE(48) TB_BAND_CASE(56)
    TB_BAND_CASE(64) {}
#undef TB_BAND_CASE
    TB_LAUNCH_CHECK("band_sort_kernel");
  } else {
    hipLaunchKernelGGL(band_kernel, dim3((unsigned)((nitems + BAND_WAVES - 1) / BAND_WAVES)), dim3(64 * BAND_WAVES),
                       lds, s, pden, B, Tf, V, rows_out, hist_in, nhist, dmin, dmax, med, sd);
    TB_LAUNCH_CHECK("band_kernel");
  }
  int rc = band_smear_finish(sd, med, lower, upper, R, Tf, V, rows_out, smear, direct_rows, band_width, s);
  if (rc) return rc;
  if (hist_out) {
    hipLaunchKernelGGL(hist_update_kernel, dim3(grid_for((long)(Tf - 1) * Tf * V)), dim3(256), 0, s, pden, B, Tf, V,
                       hist_in, hist_out);
    TB_LAUNCH_CHECK("hist_update_kernel");
  }
  tb_prof_end(TB_K_BAND, s);
  return TB_OK;
}

// tb_dimband_update: tb_band_update_part for predictions that are still to be formed -- pred[w][f][:] = act(hid[w][f][:] . W^T
// + bias), hid (B, Tf, K) the generator's last hidden buffer, W (V, K) -- without writing them to memory (dimband_kernel).
// Same outputs, bit for bit, as the linear layer followed by tb_band_update_part on its result; the band history is not
// touched (the caller keeps the last Tf-1 windows' predictions for the next batch).  Supported: forecast_len <= 48 (the eight
// waves' forecast patches have to fit the LDS),
// K = 32 or 64, act 0 / 1 (tanh) / 2 (sigmoid); anything else returns TB_ERR_UNSUPPORTED before any launch.
int tb_dimband_update(const float* hid, int B, int Tf, int K, const float* W, const float* bias, int act, int V,
                      const float* hist_in, int nhist, const double* dmin, const double* dmax, double band_width, double* med,
                      double* sd, double* lower, double* upper, int rows_out, int smear, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0) return TB_OK;
  if (Tf < 2) return tb_set_error(TB_ERR_ARG, "dimband_update: forecast_len must be >= 2 (the reference divides by Tf-1)");
  if (Tf > 48 || (K != 32 && K != 64) || act < 0 || act > 2)
    return tb_set_error(TB_ERR_UNSUPPORTED, "dimband_update: forecast_len %d / hidden %d / act %d not supported", Tf, K, act);
  if (!hid || !W || !med || !sd || !lower || !upper || V < 1) return tb_set_error(TB_ERR_ARG, "dimband_update: bad pointer");
  if ((reinterpret_cast<uintptr_t>(hid) | reinterpret_cast<uintptr_t>(W)) & 15)
    return tb_set_error(TB_ERR_ARG, "dimband_update: hid and W must be 16-byte aligned");
  if (nhist < 0 || nhist > Tf - 1 || (nhist > 0 && !hist_in)) return tb_set_error(TB_ERR_ARG, "dimband_update: nhist out of range");
  if ((dmin == nullptr) != (dmax == nullptr)) return tb_set_error(TB_ERR_ARG, "dimband_update: dmin and dmax go together");
  if ((long)B * Tf * K * 4 > 0x7FFFFFF0l) return tb_set_error(TB_ERR_UNSUPPORTED, "dimband_update: hidden buffer over 2 GiB");
  const int R = B + Tf - 1;
  if (rows_out < 1 || rows_out > R) return tb_set_error(TB_ERR_ARG, "dimband_update: rows_out %d outside [1, %d]", rows_out, R);
  if (smear && rows_out != R) return tb_set_error(TB_ERR_ARG, "dimband_update: the std smear needs every row (rows_out == B+Tf-1)");
  const int direct_rows = smear ? (R - Tf > 0 ? R - Tf : 0) : rows_out;
  const int ny = (V + 511) / 512;
  int gx = tb_num_cus() / ny;
  if (gx < 1) gx = 1;
  if (gx > rows_out) gx = rows_out;
  const dim3 g((unsigned)gx, (unsigned)ny);
  const int NS = Tf <= 16 ? 16 : (Tf <= 32 ? 32 : 48);
  const size_t lds = (size_t)2 * 2 * NS * (K + 8) * sizeof(unsigned short) + (size_t)8 * NS * (64 + 4) * sizeof(float);
  tb_prof_begin(TB_K_BAND, s);
  int rc = TB_OK;
#define TB_DIMBAND_CASE(ns, kd)                                                                                                   \
  if (NS == ns && K == kd)                                                                                                        \
    rc = launch_dimband<ns, kd>(g, lds, s, hid, B, Tf, W, bias, act, V, rows_out, hist_in, nhist, dmin, dmax, med, sd, direct_rows, \
                                band_width, lower, upper);
  TB_DIMBAND_CASE(16, 32) TB_DIMBAND_CASE(32, 32) TB_DIMBAND_CASE(48, 32)
  TB_DIMBAND_CASE(16, 64) TB_DIMBAND_CASE(32, 64) TB_DIMBAND_CASE(48, 64)
#undef TB_DIMBAND_CASE
  if (rc) return rc;
  rc = band_smear_finish(sd, med, lower, upper, R, Tf, V, rows_out, smear, direct_rows, band_width, s);
  tb_prof_end(TB_K_BAND, s);
  return rc;
}

int tb_band_update(const float* pden, int B, int Tf, int V, const float* hist_in, int nhist, float* hist_out,
                   const double* dmin, const double* dmax, double band_width, double* med, double* sd, double* lower,
                   double* upper, void* stream) {
  if (B > 0 && !hist_out) return tb_set_error(TB_ERR_ARG, "band_update: bad pointer");
  return tb_band_update_part(pden, B, Tf, V, hist_in, nhist, hist_out, dmin, dmax, band_width, med, sd, lower, upper,
                             B + Tf - 1, 1, stream);
}

int tb_adjust_scan(int variant, const double* seed, const double* reals, const double* preds, const int* masks,
                   const double* lower, const double* upper, int n, int V, double missing_gamma,
                   double anomaly_gamma, double* target, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0 || V <= 0) return TB_OK;
  if (variant < 0 || variant > 2) return tb_set_error(TB_ERR_ARG, "adjust_scan: variant must be 0, 1 or 2");
  if (!seed || !reals || !preds || !masks || !lower || !upper || !target)
    return tb_set_error(TB_ERR_ARG, "adjust_scan: null pointer");
  if (n <= ADJ_CHUNK) {   // short: the sequential form as it is
    hipLaunchKernelGGL(adjust_kernel, dim3((V + 63) / 64), dim3(64), 0, s, -1, variant, seed, reals, preds, masks,
                       lower, upper, n, V, missing_gamma, anomaly_gamma, target);
    TB_LAUNCH_CHECK("adjust_kernel");
    return TB_OK;
  }
  static std::atomic<unsigned> next_slot{0};
  const int slot = (int)(next_slot.fetch_add(1) % ADJ_FLAGS);
  hipLaunchKernelGGL(adjust_flag_reset_kernel, dim3(1), dim3(1), 0, s, slot);
  hipLaunchKernelGGL(adjust_flag_kernel, dim3(grid_for((long)n * V)), dim3(256), 0, s, slot, seed, reals, preds, lower,
                     upper, (long)n * V, V);
  hipLaunchKernelGGL(adjust_chunk_kernel, dim3((V + 63) / 64, (n + ADJ_CHUNK - 1) / ADJ_CHUNK), dim3(64), 0, s, slot,
                     variant, seed, reals, preds, masks, lower, upper, n, V, missing_gamma, anomaly_gamma, target);
  hipLaunchKernelGGL(adjust_kernel, dim3((V + 63) / 64), dim3(64), 0, s, slot, variant, seed, reals, preds, masks, lower,
                     upper, n, V, missing_gamma, anomaly_gamma, target);
  TB_LAUNCH_CHECK("adjust kernels");
  return TB_OK;
}

int tb_metrics(const double* tru, const double* pred, const int* mask, long n, double* out, double* running,
               void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!tru || !pred || !mask || !out) return tb_set_error(TB_ERR_ARG, "metrics: null pointer");
  hipLaunchKernelGGL(metric_kernel, dim3(1), dim3(256), 0, s, tru, pred, mask, n, 1e-7, out, running);
  TB_LAUNCH_CHECK("metric_kernel");
  return TB_OK;
}

size_t tb_interpolate_workspace_bytes(long rows, int E) {
  const long nblocks = (rows + FILL_ROWS - 1) / FILL_ROWS;
  return (size_t)nblocks * 2 * E * sizeof(double) + 256;
}

// data (rows, E) float64 with NaN = missing, filled IN PLACE: forward fill, backward fill of the leading gap, 0
// for an empty column (interpolate, source/dataset.py:111-115).  tpos (E) int32, nullable: tpos[c] = position of
// column c among the V target columns or -1; missing (rows - row_off, V) int32, nullable: 1 where the entry of
// row r >= row_off, target column c was missing (the reference's missing_label.csv, dataset.py:94-103,133).
int tb_interpolate(double* data, long rows, int E, const int* tpos, int V, long row_off, int* missing, void* ws,
                   size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows <= 0 || E <= 0) return TB_OK;
  if (!data || !ws) return tb_set_error(TB_ERR_ARG, "interpolate: null pointer");
  if (missing && (!tpos || V <= 0 || row_off < 0)) return tb_set_error(TB_ERR_ARG, "interpolate: missing mask needs tpos, V, row_off");
  if (ws_bytes < tb_interpolate_workspace_bytes(rows, E)) return tb_set_error(TB_ERR_WORKSPACE, "interpolate: workspace too small");
  const long nblocks = (rows + FILL_ROWS - 1) / FILL_ROWS;
  if (nblocks > 0x7FFFFFFFl) return tb_set_error(TB_ERR_UNSUPPORTED, "interpolate: too many rows");
  double* first = static_cast<double*>(ws);
  double* last = first + (size_t)nblocks * E;
  const dim3 grid((unsigned)nblocks, (E + 63) / 64);
  hipLaunchKernelGGL(fill_scan_kernel, grid, dim3(64), 0, s, data, rows, E, first, last);
  TB_LAUNCH_CHECK("fill_scan_kernel");
  hipLaunchKernelGGL(fill_carry_kernel, dim3((E + 63) / 64), dim3(64), 0, s, first, last, (int)nblocks, E);
  TB_LAUNCH_CHECK("fill_carry_kernel");
  hipLaunchKernelGGL(fill_apply_kernel, grid, dim3(64), 0, s, data, rows, E, last, tpos, V, row_off, missing);
  TB_LAUNCH_CHECK("fill_apply_kernel");
  return TB_OK;
}

constexpr int kMinmaxParts = 512;

size_t tb_minmax_workspace_bytes(int E) { return (size_t)kMinmaxParts * 2 * E * sizeof(double) + 256; }

// Column min / max of data (rows, E) fp64, NaN skipped.  init != 0: start from +-inf; init == 0: fold into what
// mn / mx hold (a series streamed in row chunks).  finish != 0: apply the reference's fix-ups afterwards.
int tb_minmax(const double* data, long rows, int E, double* mn, double* mx, int init, int finish, void* ws,
              size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (E <= 0) return TB_OK;
  if (!mn || !mx || (rows > 0 && (!data || !ws))) return tb_set_error(TB_ERR_ARG, "minmax: null pointer");
  if (rows > 0) {
    if (ws_bytes < tb_minmax_workspace_bytes(E)) return tb_set_error(TB_ERR_WORKSPACE, "minmax: workspace too small");
    long parts = rows / 256;
    if (parts > kMinmaxParts) parts = kMinmaxParts;
    if (parts < 1) parts = 1;
    double* part = static_cast<double*>(ws);
    hipLaunchKernelGGL(minmax_partial_kernel, dim3((int)parts, (E + 63) / 64), dim3(256), 0, s, data, rows, E, part);
    TB_LAUNCH_CHECK("minmax_partial_kernel");
    hipLaunchKernelGGL(minmax_fold_kernel, dim3((E + 63) / 64), dim3(64), 0, s, part, (int)parts, E, mn, mx, init);
    TB_LAUNCH_CHECK("minmax_fold_kernel");
  } else if (init) {
    // no rows to fold (int(len * split_rate) == 0): mn / mx must still be defined before the finish step reads them
    hipLaunchKernelGGL(minmax_fold_kernel, dim3((E + 63) / 64), dim3(64), 0, s, static_cast<const double*>(nullptr), 0, E, mn,
                       mx, 1);
    TB_LAUNCH_CHECK("minmax_fold_kernel");
  }
  if (finish) {
    hipLaunchKernelGGL(minmax_finish_kernel, dim3((E + 63) / 64), dim3(64), 0, s, mn, mx, E);
    TB_LAUNCH_CHECK("minmax_finish_kernel");
  }
  return TB_OK;
}

int tb_normalize(const double* data, long rows, int E, const double* mn, const double* mx, float* out32, double* out64,
                 void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows <= 0 || E <= 0) return TB_OK;
  if (!data || !mn || !mx || (!out32 && !out64)) return tb_set_error(TB_ERR_ARG, "normalize: null pointer");
  hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(rows * E, 16384)), dim3(256), 0, s, data, rows * E, E, mn, mx, out32, out64);
  TB_LAUNCH_CHECK("normalize_kernel");
  return TB_OK;
}

int tb_take_columns(const double* data, long rows, int E, const int* cols, int V, double* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (rows <= 0 || V <= 0) return TB_OK;
  if (!data || !cols || !out) return tb_set_error(TB_ERR_ARG, "take_columns: null pointer");
  hipLaunchKernelGGL(take_columns_kernel, dim3(grid_for(rows * V, 16384)), dim3(256), 0, s, data, rows, E, cols, V, out);
  TB_LAUNCH_CHECK("take_columns_kernel");
  return TB_OK;
}

int tb_metric_sums(const double* tru, const double* pred, const int* mask, long n, double* sums, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!tru || !pred || !mask || !sums) return tb_set_error(TB_ERR_ARG, "metric_sums: null pointer");
  hipLaunchKernelGGL(metric_sums_kernel, dim3(1), dim3(256), 0, s, tru, pred, mask, n, 1e-7, sums);
  TB_LAUNCH_CHECK("metric_sums_kernel");
  return TB_OK;
}

int tb_metric_finish(const double* sums, int k, double* out, double* running, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (k <= 0) return TB_OK;
  if (!sums || (!out && !running)) return tb_set_error(TB_ERR_ARG, "metric_finish: null pointer");
  hipLaunchKernelGGL(metric_finish_kernel, dim3(1), dim3(64), 0, s, sums, k, out, running);
  TB_LAUNCH_CHECK("metric_finish_kernel");
  return TB_OK;
}

}  // extern "C"
