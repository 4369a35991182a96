// K4: fused GAN / reconstruction loss reductions with analytic gradients, activation backward,
// and the fused optimizers (Adam, RMSprop) over flat parameter buffers.
//
// Reference: source/loss.py:34-51 (GANLoss), :93-141 (TIMEBANDLoss.gen_loss / dis_loss /
// l1loss / l2loss / _grad_penalty) and torch.optim.Adam / RMSprop as scheduled by
// source/trainer.py:95-98,143-144.  Everything is HBM-bound streaming work: 16-byte vector
// loads where alignment allows, one pass for the sums, one pass for the gradient.  Sums are
// accumulated in double and reduced in a fixed order (block partials, then one block), so the
// result is bitwise reproducible and the batch-global gradient-penalty norm (loss.py:136-141)
// keeps ~1e-7 relative accuracy at 2M elements.
#include "common.h"

namespace {

constexpr int NSUM = 8;   // [0] sum (Dy-real)^2  [1] sum (DGx-fake)^2  [2] sum Dy  [3] sum DGx
                          // [4] sum huber(pred-true)  [5] sum (pred-true)^2

__device__ __forceinline__ double block_sum(double v, double* sh) {
  // wave reduce then cross-wave through LDS; result valid in thread 0
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += sh[w];
  return s;
}

__global__ void loss_partial_kernel(const float* __restrict__ Dy, long nDy, const float* __restrict__ DGx,
                                    long nDGx, const float* __restrict__ tru, const float* __restrict__ pred,
                                    long nP, float real_label, float fake_label, double* part) {
  __shared__ double sh[8];
  double s[NSUM] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long stride = (long)gridDim.x * blockDim.x;
  const long i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (Dy)
    for (long i = i0; i < nDy; i += stride) {
      const float d = Dy[i] - real_label;
      s[0] += (double)(d * d);
      s[2] += (double)Dy[i];
    }
  if (DGx)
    for (long i = i0; i < nDGx; i += stride) {
      const float d = DGx[i] - fake_label;
      s[1] += (double)(d * d);
      s[3] += (double)DGx[i];
    }
  if (pred)
    for (long i = i0; i < nP; i += stride) {
      const float d = pred[i] - tru[i];
      const float ad = fabsf(d);
      s[4] += (double)(ad < 1.f ? 0.5f * d * d : ad - 0.5f);
      s[5] += (double)d * (double)d;
    }
  for (int k = 0; k < NSUM; ++k) {
    const double r = block_sum(s[k], sh);
    if (threadIdx.x == 0) part[(size_t)blockIdx.x * NSUM + k] = r;
  }
}

__global__ void loss_final_kernel(const double* __restrict__ part, int nblocks, double* sums) {
  // one block, fixed order
  __shared__ double sh[8];
  for (int k = 0; k < NSUM; ++k) {
    double v = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += blockDim.x) v += part[(size_t)b * NSUM + k];
    const double r = block_sum(v, sh);
    if (threadIdx.x == 0) sums[k] = r;
  }
}

// mode 0: discriminator LSGAN   out = [total, real, fake, 0]
// mode 1: critic WGAN + GP      out = [total, real, fake, gp]
// mode 2: generator             out = [total, adv, l1, l2, gp]
// running (nullable, float[8]): D: [0]+=real [1]+=fake ; G: [2]+=adv [3]+=l1 [4]+=l2 [5]+=gp
__device__ void loss_finish_dev(const double* sums, int mode, double nD_real, double nD_fake, double nP, double batch,
                                float l1_w, float l2_w, float gp_w, float* out, float* running, float* total) {
  if (mode == 0) {
    const float real = (float)(sums[0] / nD_real), fake = (float)(sums[1] / nD_fake);
    out[0] = fake + real; out[1] = real; out[2] = fake; out[3] = 0.f;
    if (running) { running[0] += real; running[1] += fake; }
  } else if (mode == 1) {
    const float real = (float)(-sums[2] / nD_real), fake = (float)(sums[3] / nD_fake);
    const double n = sqrt(sums[5]);
    const float gp = gp_w * (float)((1.0 - n) * (1.0 - n) / batch);
    out[0] = fake + real + gp; out[1] = real; out[2] = fake; out[3] = gp;
  } else {
    const float adv = (float)(sums[1] / nD_fake);
    const float l1 = l1_w * (float)(sums[4] / nP);
    const float l2 = l2_w * (float)(sums[5] / nP);
    const double n = sqrt(sums[5]);
    const float gp = gp_w * (float)((1.0 - n) * (1.0 - n) / batch);
    out[0] = adv + l1 + l2 + gp; out[1] = adv; out[2] = l1; out[3] = l2; out[4] = gp;
    if (running) { running[2] += adv; running[3] += l1; running[4] += l2; running[5] += gp; }
  }
  if (total) total[0] = out[0];
}

__global__ void loss_finish_kernel(const double* __restrict__ sums, int mode, double nD_real, double nD_fake,
                                   double nP, double batch, float l1_w, float l2_w, float gp_w, float* out,
                                   float* running, float* total) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  loss_finish_dev(sums, mode, nD_real, nD_fake, nP, batch, l1_w, l2_w, gp_w, out, running, total);
}

// loss_final_kernel and loss_finish_kernel in one launch (tb_loss_sums_finish: no all-reduce between them)
__global__ void loss_final_finish_kernel(const double* __restrict__ part, int nblocks, double* sums, int mode,
                                         double nD_real, double nD_fake, double nP, double batch, float l1_w, float l2_w,
                                         float gp_w, float* out, float* running, float* total) {
  // 512 threads: wave k sums the partials of sum k (lanes stride the blocks, then a fixed-order tree) -- the eight sums
  // side by side instead of eight block reductions in a row
  __shared__ double fin[NSUM];
  const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
  double v = 0.0;
  for (int b = lane; b < nblocks; b += 64) v += part[(size_t)b * NSUM + k];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if (lane == 0) { sums[k] = v; fin[k] = v; }
  __syncthreads();
  if (threadIdx.x == 0) loss_finish_dev(fin, mode, nD_real, nD_fake, nP, batch, l1_w, l2_w, gp_w, out, running, total);
}

// both halves of the discriminator's real + fake output in one launch: p = [real half | fake half], n each
__global__ void gan_grad_pair_kernel(const float* __restrict__ p, long n_each, double n_real_global, double n_fake_global,
                                     int wgan, const float* __restrict__ upstream, float* __restrict__ dp) {
  const float up = upstream ? upstream[0] : 1.f;
  const float inv_r = (float)(1.0 / n_real_global), inv_f = (float)(1.0 / n_fake_global);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < 2 * n_each; i += (long)gridDim.x * blockDim.x) {
    const bool real = i < n_each;
    const float label = real ? 0.9f : 0.1f, sign = real ? -1.f : 1.f, inv = real ? inv_r : inv_f;
    dp[i] = wgan ? up * sign * inv : up * 2.f * (p[i] - label) * inv;
  }
}

// d(loss)/dD for the adversarial terms.  lsgan: up * 2 (p - label) / n ; wgan: up * sign / n
__global__ void gan_grad_kernel(const float* __restrict__ p, long n_local, double n_global, float label,
                                int wgan, float sign, const float* __restrict__ upstream, float* __restrict__ dp) {
  const float up = upstream ? upstream[0] : 1.f;
  const float inv = (float)(1.0 / n_global);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_local; i += (long)gridDim.x * blockDim.x)
    dp[i] = wgan ? up * sign * inv : up * 2.f * (p[i] - label) * inv;
}

// d/dpred of l1_w*SmoothL1 + l2_w*MSE + gp_w*(1-||d||)^2/B, d = pred - true (loss.py:127-141)
__global__ void recon_grad_kernel(const float* __restrict__ tru, const float* __restrict__ pred, long n_local,
                                  double n_global, double batch, const double* __restrict__ sums, float l1_w,
                                  float l2_w, float gp_w, const float* __restrict__ upstream,
                                  float* __restrict__ dpred, int accumulate) {
  const float up = upstream ? upstream[0] : 1.f;
  const double nrm = sqrt(sums[5]);
  const float cgp = nrm > 0.0 ? (float)(-2.0 * gp_w * (1.0 - nrm) / (batch * nrm)) : 0.f;
  const float c1 = l1_w * (float)(1.0 / n_global), c2 = l2_w * 2.f * (float)(1.0 / n_global);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_local; i += (long)gridDim.x * blockDim.x) {
    const float d = pred[i] - tru[i];
    const float cl = fminf(fmaxf(d, -1.f), 1.f);
    const float g = up * (c1 * cl + c2 * d + cgp * d);
    dpred[i] = accumulate ? dpred[i] + g : g;
  }
}

// dz = dy * act'(y)   act 1: tanh, 2: sigmoid
__global__ void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dz,
                               long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = y[i];
    dz[i] = dy[i] * (act == 1 ? (1.f - v * v) : v * (1.f - v));
  }
}

// out[c] = sum_r a[r][c] in two launches.  Block (x, y): rows slice x, columns [y*cw, (y+1)*cw); the 256
// threads are 256/cw row lanes x cw columns (cw a power of two <= 64, so narrow matrices still use
// every lane and each block-iteration reads 256 consecutive floats); double accumulation, fixed order.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ a, long rows, int cols, int cw,
                                                     double* __restrict__ part) {
  __shared__ double sh[256];
  const int tx = threadIdx.x % cw, ty = threadIdx.x / cw, rpb = 256 / cw;
  const int c = blockIdx.y * cw + tx;
  const long per = (rows + gridDim.x - 1) / gridDim.x;
  const long r0 = blockIdx.x * per, r1 = r0 + per < rows ? r0 + per : rows;
  double s = 0.0;
  if (c < cols) {
    long r = r0 + ty;
    for (; r + 3L * rpb < r1; r += 4L * rpb) {   // 4 independent loads in flight
      const float v0 = a[r * cols + c], v1 = a[(r + rpb) * cols + c], v2 = a[(r + 2L * rpb) * cols + c],
                  v3 = a[(r + 3L * rpb) * cols + c];
      s += (double)v0; s += (double)v1; s += (double)v2; s += (double)v3;
    }
    for (; r < r1; r += rpb) s += (double)a[r * cols + c];
  }
  sh[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && c < cols) {
    double t = 0.0;
    for (int k = 0; k < rpb; ++k) t += sh[k * cw + tx];
    part[(size_t)blockIdx.x * cols + c] = t;
  }
}

__global__ __launch_bounds__(64) void colsum_final_kernel(const double* __restrict__ part, int nparts, int cols,
                                                          float* out, int accumulate) {
  // one wave per column: lanes stride the partials, then a fixed-order tree over the lanes
  const int c = blockIdx.x, lane = threadIdx.x;
  double s = 0.0;
  for (int p = lane; p < nparts; p += 64) s += part[(size_t)p * cols + c];
  __shared__ double sh[64];
  sh[lane] = s;
  __syncthreads();
  for (int w = 32; w > 0; w >>= 1) {
    if (lane < w) sh[lane] += sh[lane + w];
    __syncthreads();
  }
  if (lane == 0) out[c] = accumulate ? out[c] + (float)sh[0] : (float)sh[0];
}

// Tail of a linear layer's backward in ONE launch (tb_linear_bwd): the split-K slabs of dW = dz^T.x summed in a fixed
// order (blocks < dw_blocks: 8 lanes per element, lane l takes slabs l, l + 8, ... four loads in flight, then a butterfly
// over the 8 lanes -- a one-tile product over K = 131072 comes in 512 slabs) and the column-sum partials of db summed as
// colsum_final_kernel does (one block per column behind them, its first wave).  Either part may be empty.
__global__ __launch_bounds__(256) void linear_finish_kernel(TbGemmDeferred d, int dw_blocks, const double* __restrict__ part,
                                                            int nparts, int cols, float* db, int accumulate) {
  if ((int)blockIdx.x < dw_blocks) {
    const long n = (long)d.M * d.N, i = (long)blockIdx.x * 256 + threadIdx.x, e = i >> 3;
    const int l = (int)(i & 7);
    float s = 0.f;
    if (e < n) {
      int z = l;
      for (; z + 24 < d.splits; z += 32) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = d.slab[(size_t)(z + 8 * k) * n + e];
#pragma unroll
        for (int k = 0; k < 4; ++k) s += v[k];
      }
      for (; z < d.splits; z += 8) s += d.slab[(size_t)z * n + e];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (e < n && l == 0) {
      float* c = d.C + (e / d.N) * d.ldc + e % d.N;
      *c = d.accumulate ? *c + s : s;
    }
    return;
  }
  const int lane = threadIdx.x;
  if (lane >= 64) return;
  const int c = blockIdx.x - dw_blocks;
  double s = 0.0;
  for (int p = lane; p < nparts; p += 64) s += part[(size_t)p * cols + c];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);   // one wave: a fixed-order tree
  if (lane == 0) db[c] = accumulate ? db[c] + (float)s : (float)s;
}

// torch.optim.Adam (no weight decay / amsgrad); operation order follows torch's single-tensor path
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long n, float one_minus_b1, float b2, float one_minus_b2,
                            float step_size, float bc2_sqrt, float eps) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float mi = m[i] + (gi - m[i]) * one_minus_b1;
    const float vi = v[i] * b2 + one_minus_b2 * gi * gi;
    m[i] = mi;
    v[i] = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (mi / denom);
  }
}

// torch.optim.RMSprop (no momentum, not centered).  fresh != 0: square_avg starts at zero and
// is not read (the reference re-creates the optimizer for every batch, trainer.py:143-144)
__global__ void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ sq, long n,
                               float lr, float alpha, float eps, int fresh) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gi = g[i];
    const float s = (fresh ? 0.f : sq[i] * alpha) + (1.f - alpha) * gi * gi;
    if (sq) sq[i] = s;
    p[i] = p[i] - lr * (gi / (sqrtf(s) + eps));
  }
}

inline int grid_for(long n, int cap = 2048) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return (int)b;
}

}  // namespace

extern "C" {

size_t tb_loss_workspace_bytes(void) { return (size_t)1024 * NSUM * sizeof(double) + 256; }

// Stage 1 of every loss: the sums.  `sums` is a device double[8]; in data-parallel runs the host
// all-reduces it (sum) between this call and the finish/grad calls.
int tb_loss_sums(const float* Dy, long nDy, const float* DGx, long nDGx, const float* tru, const float* pred,
                 long nP, double* sums, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!sums || !ws) return tb_set_error(TB_ERR_ARG, "loss_sums: null pointer");
  if (ws_bytes < tb_loss_workspace_bytes()) return tb_set_error(TB_ERR_WORKSPACE, "loss_sums: workspace too small");
  long nmax = nDy > nDGx ? nDy : nDGx;
  if (nP > nmax) nmax = nP;
  const int blocks = grid_for(nmax, 1024);
  double* part = static_cast<double*>(ws);
  tb_prof_begin(TB_K_LOSS, s);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, s, Dy, nDy, DGx, nDGx, tru, pred, nP, 0.9f,
                     0.1f, part);
  TB_LAUNCH_CHECK("loss_partial_kernel");
  hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, part, blocks, sums);
  TB_LAUNCH_CHECK("loss_final_kernel");
  tb_prof_end(TB_K_LOSS, s);
  return TB_OK;
}

int tb_loss_finish(const double* sums, int mode, double nD_real, double nD_fake, double nP, double batch,
                   float l1_w, float l2_w, float gp_w, float* out, float* running, float* total, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!sums || !out) return tb_set_error(TB_ERR_ARG, "loss_finish: null pointer");
  hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, s, sums, mode, nD_real, nD_fake, nP, batch, l1_w,
                     l2_w, gp_w, out, running, total);
  TB_LAUNCH_CHECK("loss_finish_kernel");
  return TB_OK;
}

int tb_loss_sums_finish(const float* Dy, long nDy, const float* DGx, long nDGx, const float* tru, const float* pred, long nP,
                        double* sums, void* ws, size_t ws_bytes, int mode, double nD_real, double nD_fake, double nP_global,
                        double batch, float l1_w, float l2_w, float gp_w, float* out, float* running, float* total,
                        void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!sums || !ws || !out) return tb_set_error(TB_ERR_ARG, "loss_sums_finish: null pointer");
  if (ws_bytes < tb_loss_workspace_bytes()) return tb_set_error(TB_ERR_WORKSPACE, "loss_sums_finish: workspace too small");
  long nmax = nDy > nDGx ? nDy : nDGx;
  if (nP > nmax) nmax = nP;
  const int blocks = grid_for(nmax, 1024);
  double* part = static_cast<double*>(ws);
  tb_prof_begin(TB_K_LOSS, s);
  hipLaunchKernelGGL(loss_partial_kernel, dim3(blocks), dim3(256), 0, s, Dy, nDy, DGx, nDGx, tru, pred, nP, 0.9f,
                     0.1f, part);
  TB_LAUNCH_CHECK("loss_partial_kernel");
  static_assert(NSUM == 8, "loss_final_finish_kernel: one wave per sum");
  hipLaunchKernelGGL(loss_final_finish_kernel, dim3(1), dim3(512), 0, s, part, blocks, sums, mode, nD_real, nD_fake,
                     nP_global, batch, l1_w, l2_w, gp_w, out, running, total);
  TB_LAUNCH_CHECK("loss_final_finish_kernel");
  tb_prof_end(TB_K_LOSS, s);
  return TB_OK;
}

int tb_gan_loss_grad_pair(const float* p, long n_each, double n_real_global, double n_fake_global, int wgan,
                          const float* upstream, float* dp, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_each <= 0) return TB_OK;
  if (!dp || (!wgan && !p)) return tb_set_error(TB_ERR_ARG, "gan_loss_grad_pair: null pointer");
  hipLaunchKernelGGL(gan_grad_pair_kernel, dim3(grid_for(2 * n_each)), dim3(256), 0, s, p, n_each, n_real_global,
                     n_fake_global, wgan, upstream, dp);
  TB_LAUNCH_CHECK("gan_grad_pair_kernel");
  return TB_OK;
}

int tb_gan_loss_grad(const float* p, long n_local, double n_global, int target_is_real, int wgan,
                     const float* upstream, float* dp, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_local <= 0) return TB_OK;
  if (!dp || (!wgan && !p)) return tb_set_error(TB_ERR_ARG, "gan_loss_grad: null pointer");
  const float label = target_is_real ? 0.9f : 0.1f;
  const float sign = target_is_real ? -1.f : 1.f;
  hipLaunchKernelGGL(gan_grad_kernel, dim3(grid_for(n_local)), dim3(256), 0, s, p, n_local, n_global, label, wgan,
                     sign, upstream, dp);
  TB_LAUNCH_CHECK("gan_grad_kernel");
  return TB_OK;
}

int tb_recon_loss_grad(const float* tru, const float* pred, long n_local, double n_global, double batch,
                       const double* sums, float l1_w, float l2_w, float gp_w, const float* upstream,
                       float* dpred, int accumulate, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n_local <= 0) return TB_OK;
  if (!tru || !pred || !sums || !dpred) return tb_set_error(TB_ERR_ARG, "recon_loss_grad: null pointer");
  hipLaunchKernelGGL(recon_grad_kernel, dim3(grid_for(n_local)), dim3(256), 0, s, tru, pred, n_local, n_global,
                     batch, sums, l1_w, l2_w, gp_w, upstream, dpred, accumulate);
  TB_LAUNCH_CHECK("recon_grad_kernel");
  return TB_OK;
}

int tb_act_bwd(const float* dy, const float* y, float* dz, long n, int act, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0) return TB_OK;
  if (act != 1 && act != 2) return tb_set_error(TB_ERR_ARG, "act_bwd: act must be 1 (tanh) or 2 (sigmoid)");
  hipLaunchKernelGGL(act_bwd_kernel, dim3(grid_for(n)), dim3(256), 0, s, dy, y, dz, n, act);
  TB_LAUNCH_CHECK("act_bwd_kernel");
  return TB_OK;
}

constexpr int kColsumMaxParts = 1024;

size_t tb_colsum_workspace_bytes(int cols) { return (size_t)kColsumMaxParts * cols * sizeof(double) + 256; }

int tb_colsum_acc(const float* a, long rows, int cols, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream);

int tb_colsum(const float* a, long rows, int cols, float* out, void* ws, size_t ws_bytes, void* stream) {
  return tb_colsum_acc(a, rows, cols, out, 0, ws, ws_bytes, stream);
}

// accumulate: out[c] += the column sum (internal form used by tb_linear_bwd)
int tb_colsum_acc(const float* a, long rows, int cols, float* out, int accumulate, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (cols <= 0) return TB_OK;
  if (!a || !out || !ws) return tb_set_error(TB_ERR_ARG, "colsum: null pointer");
  if (ws_bytes < tb_colsum_workspace_bytes(cols)) return tb_set_error(TB_ERR_WORKSPACE, "colsum: workspace too small");
  int cw = 1;
  while (cw < cols && cw < 64) cw *= 2;
  const int ycount = (cols + cw - 1) / cw;
  // enough blocks to fill the chip, each with at least ~64 block-iterations of rows
  long parts = rows / (64L * (256 / cw));
  const long want = (4L * tb_num_cus() + ycount - 1) / ycount;
  if (parts > want) parts = want;
  if (parts > kColsumMaxParts) parts = kColsumMaxParts;
  if (parts < 1) parts = 1;
  double* part = static_cast<double*>(ws);
  hipLaunchKernelGGL(colsum_kernel, dim3((int)parts, ycount), dim3(256), 0, s, a, rows, cols, cw, part);
  TB_LAUNCH_CHECK("colsum_kernel");
  hipLaunchKernelGGL(colsum_final_kernel, dim3(cols), dim3(64), 0, s, part, (int)parts, cols, out, accumulate);
  TB_LAUNCH_CHECK("colsum_final_kernel");
  return TB_OK;
}

}  // extern "C"

// The partial pass of the column sum alone (tb_linear_bwd sums the partials in its finish kernel).
int tb_colsum_partials(const float* a, long rows, int cols, void* ws, size_t ws_bytes, hipStream_t s, int* nparts) {
  if (!a || !ws) return tb_set_error(TB_ERR_ARG, "colsum: null pointer");
  if (ws_bytes < tb_colsum_workspace_bytes(cols)) return tb_set_error(TB_ERR_WORKSPACE, "colsum: workspace too small");
  int cw = 1;
  while (cw < cols && cw < 64) cw *= 2;
  const int ycount = (cols + cw - 1) / cw;
  long parts = rows / (64L * (256 / cw));
  const long want = (4L * tb_num_cus() + ycount - 1) / ycount;
  if (parts > want) parts = want;
  if (parts > kColsumMaxParts) parts = kColsumMaxParts;
  if (parts < 1) parts = 1;
  hipLaunchKernelGGL(colsum_kernel, dim3((int)parts, ycount), dim3(256), 0, s, a, rows, cols, cw, static_cast<double*>(ws));
  TB_LAUNCH_CHECK("colsum_kernel");
  *nparts = (int)parts;
  return TB_OK;
}

// d: a deferred dW product (d->splits == 0: nothing pending); part / nparts / cols / db: column-sum partials (db NULL: none)
int tb_linear_finish(const TbGemmDeferred* d, const double* part, int nparts, int cols, float* db, int accumulate, hipStream_t s) {
  const long n = d && d->splits ? (long)d->M * d->N : 0;
  const int dw_blocks = (int)((n * 8 + 255) / 256), col_blocks = db ? cols : 0;
  if (dw_blocks + col_blocks == 0) return TB_OK;
  TbGemmDeferred dd = {};
  if (n) dd = *d;
  hipLaunchKernelGGL(linear_finish_kernel, dim3(dw_blocks + col_blocks), dim3(256), 0, s, dd, dw_blocks, part, nparts, cols, db, accumulate);
  TB_LAUNCH_CHECK("linear_finish_kernel");
  return TB_OK;
}

namespace {
// grad += grad2; grad2 = 0  (16 bytes per lane; the tail element-wise)
__global__ void grad_merge_kernel(float* __restrict__ g, float* __restrict__ g2, long n) {
  const long n4 = n / 4;
  f32x4* a = reinterpret_cast<f32x4*>(g);
  f32x4* b = reinterpret_cast<f32x4*>(g2);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    a[i] = a[i] + b[i];
    b[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    g[i] += g2[i];
    g2[i] = 0.f;
  }
}
// out = [a ; b + b2] (floats): the upstream gradient of the joint generator backward -- the discriminator phase's dfake, then
// the generator phase's two contributions (reconstruction / gradient-penalty terms and the path through D), summed on the way
__global__ void concat2_sum_kernel(const float* __restrict__ a, long na, const float* __restrict__ b, const float* __restrict__ b2,
                                   long nb, float* __restrict__ out, int vec) {
  const long stride = (long)gridDim.x * blockDim.x, i0 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (vec) {
    const long na4 = na / 4, nb4 = nb / 4;
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a);
    const f32x4* b4 = reinterpret_cast<const f32x4*>(b);
    const f32x4* c4 = reinterpret_cast<const f32x4*>(b2);
    f32x4* o4 = reinterpret_cast<f32x4*>(out);
    for (long i = i0; i < na4 + nb4; i += stride) o4[i] = i < na4 ? a4[i] : b4[i - na4] + c4[i - na4];
  } else {
    for (long i = i0; i < na + nb; i += stride) out[i] = i < na ? a[i] : b[i - na] + b2[i - na];
  }
}
}  // namespace

extern "C" {

int tb_zero(void* p, size_t bytes, void* stream) {
  if (!bytes) return TB_OK;
  if (!p) return tb_set_error(TB_ERR_ARG, "zero: null pointer");
  return tb_check_hip(hipMemsetAsync(p, 0, bytes, static_cast<hipStream_t>(stream)), "zero");
}

int tb_grad_merge(float* g, float* g2, long n, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0) return TB_OK;
  if (!g || !g2 || ((reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(g2)) & 15))
    return tb_set_error(TB_ERR_ARG, "grad_merge: null or unaligned pointer");
  hipLaunchKernelGGL(grad_merge_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, g, g2, n);
  TB_LAUNCH_CHECK("grad_merge_kernel");
  return TB_OK;
}

int tb_concat2(const void* a, size_t bytes_a, const void* b, size_t bytes_b, void* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if ((bytes_a && !a) || (bytes_b && !b) || ((bytes_a || bytes_b) && !out)) return tb_set_error(TB_ERR_ARG, "concat2: null pointer");
  if (bytes_a) {
    int rc = tb_check_hip(hipMemcpyAsync(out, a, bytes_a, hipMemcpyDeviceToDevice, s), "concat2");
    if (rc) return rc;
  }
  if (bytes_b) return tb_check_hip(hipMemcpyAsync(static_cast<char*>(out) + bytes_a, b, bytes_b, hipMemcpyDeviceToDevice, s), "concat2");
  return TB_OK;
}

int tb_concat2_sum(const float* a, long na, const float* b, const float* b2, long nb, float* out, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (na < 0 || nb < 0) return tb_set_error(TB_ERR_ARG, "concat2_sum: negative length");
  if (na + nb == 0) return TB_OK;
  if ((na && !a) || (nb && (!b || !b2)) || !out) return tb_set_error(TB_ERR_ARG, "concat2_sum: null pointer");
  const uintptr_t bits = reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(b2) |
                         reinterpret_cast<uintptr_t>(out);
  const int vec = (bits & 15) == 0 && (na & 3) == 0 && (nb & 3) == 0;
  hipLaunchKernelGGL(concat2_sum_kernel, dim3(grid_for(vec ? (na + nb) / 4 : na + nb)), dim3(256), 0, s, a, na, b, b2, nb, out, vec);
  TB_LAUNCH_CHECK("concat2_sum_kernel");
  return TB_OK;
}

int tb_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                 double eps, int step, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0) return TB_OK;
  if (!p || !g || !m || !v || step < 1) return tb_set_error(TB_ERR_ARG, "adam_step: bad argument");
  const double bc1 = 1.0 - pow(b1, (double)step), bc2 = 1.0 - pow(b2, (double)step);
  tb_prof_begin(TB_K_OPTIM, s);
  hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, m, v, n, (float)(1.0 - b1), (float)b2,
                     (float)(1.0 - b2), (float)(lr / bc1), (float)sqrt(bc2), (float)eps);
  TB_LAUNCH_CHECK("adam_kernel");
  tb_prof_end(TB_K_OPTIM, s);
  return TB_OK;
}

int tb_rmsprop_step(float* p, const float* g, float* sq, long n, double lr, double alpha, double eps, int fresh,
                    void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (n <= 0) return TB_OK;
  if (!p || !g || (!fresh && !sq)) return tb_set_error(TB_ERR_ARG, "rmsprop_step: bad argument");
  tb_prof_begin(TB_K_OPTIM, s);
  hipLaunchKernelGGL(rmsprop_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, g, sq, n, (float)lr, (float)alpha,
                     (float)eps, fresh);
  TB_LAUNCH_CHECK("rmsprop_kernel");
  tb_prof_end(TB_K_OPTIM, s);
  return TB_OK;
}

}  // extern "C"
