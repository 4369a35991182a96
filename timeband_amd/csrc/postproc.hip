// K5 window gather, K6 band (median / std over horizons), K7 adjust (imputation / anomaly scan),
// denormalise and the three scores -- the data side of the hot path.
//
// Reference: source/dataset.py:199-214 (windowing), :229-242 (denormalize), source/utils/dataset.py:17-23;
// source/trainer.py:238-287, source/cleaner.py:130-179, source/predictor.py:106-155 (adjust / predicts);
// source/metric.py:30-73.  Post-processing is float64 like the reference's numpy code; every
// multiply-add that numpy performs as two rounded operations is kept as two (no fma contraction).
#include "common.h"

#pragma clang fp contract(off)

namespace {

// ---- K5 ----------------------------------------------------------------------------------
// window w: i = starts ? starts[w] : first + w*stride;
//   enc[w] = series[i-To : i, :]            (To*E contiguous floats)
//   dec[w] = series[i : i+Tf, target_cols]
__global__ void window_gather_kernel(const float* __restrict__ series, int E, const long* __restrict__ starts,
                                     long first, long stride, int nwin, int To, int Tf,
                                     const int* __restrict__ tcols, int V, float* __restrict__ enc,
                                     float* __restrict__ dec) {
  const long ne = (long)To * E, nd = (long)Tf * V;
  const long per = ne + nd;
  const long total = per * nwin;
  for (long g = blockIdx.x * (long)blockDim.x + threadIdx.x; g < total; g += (long)gridDim.x * blockDim.x) {
    const int w = (int)(g / per);
    const long r = g - (long)w * per;
    const long i = starts ? starts[w] : first + (long)w * stride;
    if (r < ne) {
      enc[(long)w * ne + r] = series[(i - To) * E + r];
    } else {
      const long k = r - ne;
      const int f = (int)(k / V), v = (int)(k % V);
      dec[(long)w * nd + k] = series[(i + f) * E + tcols[v]];
    }
  }
}

// ---- denormalise (dataset.py:229-242) ----------------------------------------------------------
// fp32 0.5*(x+1); times fp64 (max-min); plus fp64 min; rounded back to fp32
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

// ---- K6 band -------------------------------------------------------------------------------
// One wave per (row r, variable v).  Row r (time offset from the batch's first forecast step)
// collects the predictions made for it at every horizon f by window w = r - f: from this batch
// (0 <= w < B) or from the history of the last Tf-1 windows (w < 0).  nanmedian / nanstd (ddof 0).
constexpr int BAND_WAVES = 4;

__global__ __launch_bounds__(64 * BAND_WAVES) void band_kernel(const float* __restrict__ pden, int B, int Tf, int V,
                                                               const float* __restrict__ hist, int nhist,
                                                               double* __restrict__ med, double* __restrict__ sd) {
  extern __shared__ float vals_all[];   // [BAND_WAVES][Tf]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long item = (long)blockIdx.x * BAND_WAVES + wave;
  const long nitems = (long)(B + Tf - 1) * V;
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

// std smear over the last Tf rows (trainer.py:280-282), then the band
__global__ void band_finish_kernel(const double* __restrict__ med, double* __restrict__ sd, int R, int Tf, int V,
                                   double width, double* __restrict__ lower, double* __restrict__ upper) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  for (int f = Tf - 1; f >= 1; --f) {
    const double gamma = (double)(Tf - f) / (double)(Tf - 1);
    const double add = sd[(long)(R - f - 1) * V + v] * gamma;
    sd[(long)(R - f) * V + v] = sd[(long)(R - f) * V + v] + add;
  }
  for (int r = 0; r < R; ++r) {
    const double m = med[(long)r * V + v], w = width * sd[(long)r * V + v];
    lower[(long)r * V + v] = m - w;
    upper[(long)r * V + v] = m + w;
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
// Sequential in p (each row uses the already-adjusted previous row), parallel over variables.
__global__ void adjust_kernel(int variant, const double* __restrict__ seed, const double* __restrict__ reals,
                              const double* __restrict__ preds, const int* __restrict__ masks,
                              const double* __restrict__ lower, const double* __restrict__ upper, int n, int V,
                              double a, double b, double* __restrict__ target) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= V) return;
  double prev = seed[v];
  for (int p = 0; p < n; ++p) {
    const long o = (long)p * V + v;
    double value = reals[o];
    const double m = (double)masks[o];
    const double pr = preds[o], lo = lower[o], up = upper[o];
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
    target[o] = value;
    prev = value;
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
  const long total = ((long)To * E + (long)Tf * V) * nwin;
  tb_prof_begin(TB_K_GATHER, s);
  hipLaunchKernelGGL(window_gather_kernel, dim3(grid_for(total)), dim3(256), 0, s, series, E, starts, first, stride,
                     nwin, To, Tf, target_cols, V, enc, dec);
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

// pden: (B,Tf,V) denormalised predictions of this batch.  hist_in/hist_out: (Tf-1,Tf,V) predictions of
// the last Tf-1 windows before / after this batch (distinct buffers), nhist = valid windows in hist_in.
// med/sd/lower/upper: (B+Tf-1, V) float64 outputs (sd is returned AFTER the smear).
int tb_band_update(const float* pden, int B, int Tf, int V, const float* hist_in, int nhist, float* hist_out,
                   double band_width, double* med, double* sd, double* lower, double* upper, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0) return TB_OK;
  if (Tf < 2) return tb_set_error(TB_ERR_ARG, "band_update: forecast_len must be >= 2 (the reference divides by Tf-1)");
  if (!pden || !hist_in || !hist_out || !med || !sd || !lower || !upper || hist_in == hist_out)
    return tb_set_error(TB_ERR_ARG, "band_update: bad pointer");
  if (nhist < 0 || nhist > Tf - 1) return tb_set_error(TB_ERR_ARG, "band_update: nhist out of range");
  const int R = B + Tf - 1;
  const long nitems = (long)R * V;
  const size_t lds = (size_t)BAND_WAVES * Tf * sizeof(float);
  if (lds > 60 * 1024) return tb_set_error(TB_ERR_UNSUPPORTED, "band_update: forecast_len %d too long", Tf);
  tb_prof_begin(TB_K_BAND, s);
  hipLaunchKernelGGL(band_kernel, dim3((unsigned)((nitems + BAND_WAVES - 1) / BAND_WAVES)), dim3(64 * BAND_WAVES),
                     lds, s, pden, B, Tf, V, hist_in, nhist, med, sd);
  TB_LAUNCH_CHECK("band_kernel");
  hipLaunchKernelGGL(band_finish_kernel, dim3((V + 63) / 64), dim3(64), 0, s, med, sd, R, Tf, V, band_width, lower,
                     upper);
  TB_LAUNCH_CHECK("band_finish_kernel");
  hipLaunchKernelGGL(hist_update_kernel, dim3(grid_for((long)(Tf - 1) * Tf * V)), dim3(256), 0, s, pden, B, Tf, V,
                     hist_in, hist_out);
  TB_LAUNCH_CHECK("hist_update_kernel");
  tb_prof_end(TB_K_BAND, s);
  return TB_OK;
}

int tb_adjust_scan(int variant, const double* seed, const double* reals, const double* preds, const int* masks,
                   const double* lower, const double* upper, int n, int V, double missing_gamma,
                   double anomaly_gamma, double* target, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0 || V <= 0) return TB_OK;
  if (variant < 0 || variant > 2) return tb_set_error(TB_ERR_ARG, "adjust_scan: variant must be 0, 1 or 2");
  if (!seed || !reals || !preds || !masks || !lower || !upper || !target)
    return tb_set_error(TB_ERR_ARG, "adjust_scan: null pointer");
  hipLaunchKernelGGL(adjust_kernel, dim3((V + 63) / 64), dim3(64), 0, s, variant, seed, reals, preds, masks, lower,
                     upper, n, V, missing_gamma, anomaly_gamma, target);
  TB_LAUNCH_CHECK("adjust_kernel");
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

}  // extern "C"
