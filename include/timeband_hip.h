/* timeband_hip.h -- C-ABI of libtimeband_hip.so: the gfx950 (MI355X) kernels behind Timeband's
 * LSTM-GAN training / cleaning hot path.
 *
 * The reference (handal95/Timeband) has no FFI of its own: its hot path is torch library ops called
 * from Python.  Each entry point below names the reference call site (file:line under
 * /root/reference) whose torch op(s) it replaces.  Conventions:
 *   - every pointer is a DEVICE pointer unless the comment says "host";
 *   - tensors are dense, row-major, float32 unless typed otherwise;
 *   - `stream` is a hipStream_t (void*); work is enqueued, never synchronised;
 *   - the library allocates nothing on the device: callers pass workspaces sized by the
 *     matching *_workspace_bytes() query, 256-byte aligned, contents arbitrary (nothing has to be
 *     cleared); calls that may run at the same time (different streams) need different workspaces;
 *   - return value 0 = ok; otherwise tb_last_error() (host string, thread-local) says why.
 */
#ifndef TIMEBAND_HIP_H
#define TIMEBAND_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- library ------------------------------------------------------------------------------ */
int tb_version(void);
const char* tb_last_error(void);
/* Select `device`, check it is gfx950, cache its CU count.  Mirrors init_device
 * (source/utils/initiate.py:21-28), one process per GPU. */
int tb_init(int device);
int tb_device_cus(void);
/* HIP-event timing of one kernel family (0 gemm, 1 lstm_fwd, 2 lstm_bwd, 3 loss, 4 optim,
 * 5 gather, 6 band) on the stream it runs on; read after a device synchronise.  algo_bytes = the
 * SURVEY.md 8(d) algorithmic bytes of the timed launches (LSTM kernels only). */
int tb_prof_enable(int kernel_id, int on);
int tb_prof_read(int kernel_id, int* count_host, double* total_ms_host, double* algo_bytes_host);

/* ---- LSTM layer (torch.nn.LSTM, one layer, batch_first, zero initial state) --------------------
 * Reference call sites: source/utils/lstm_layer.py:31-33,44-46 (generator lstm0/1/2) and :75,83
 * (discriminator); Gen-B twins source/models.py:44-46,64-66,101,111.
 *   x (B,T,in)  w_ih (4H,in)  w_hh (4H,H)  b_ih,b_hh (4H)  gate order i,f,g,o
 * forward : y (B,T,H).  `gates` (B,T,4H) receives the activated gates and `cstash` (B,T,H) the
 *           cell states when train != 0 (the BPTT stash); with train == 0 `gates` is scratch and
 *           cstash may be NULL.  For in_dim <= 64, a multiple of 4 (<= 32 at H = 256), the input
 *           projection is computed inside the recurrent kernel: `gates` is then output only and may be
 *           NULL with train == 0 (TB_FUSE_PROJ=0 in the environment: projection by GEMM for every shape).
 * backward: dy (B,T,H) -> dx (B,T,in) [nullable], dw_ih, dw_hh, db_ih, db_hh [each nullable];
 *           `gates` is consumed (overwritten with the pre-activation gradients -- except when all four parameter-gradient
 *           pointers are null and dx (in <= 32, hidden 128 / 256) is formed inside the kernel: nothing reads them then, and
 *           the buffer keeps the activated gates).
 *           accumulate != 0: the four parameter gradients are ADDED to what the buffers hold (what
 *           autograd's `.grad +=` does, trainer.py:179-180,193); dx is always overwritten.  The same flag
 *           on tb_linear_bwd (dW, db) and tb_head_bwd (dw, db).
 * H must be one of 16,32,64,128,256,512 (tb_lstm_supported).  `err` is a device int[4]: err[0]
 * becomes non-zero if an inter-workgroup wait timed out (results are then invalid). */
int tb_lstm_supported(int H);
/* 1 if this shape's launch is EXCLUSIVE: its workgroups leave no room for a second clustered LSTM workgroup on a CU
 * (BPTT at H = 256 with B > 512 on 256 CUs), so no other tb_lstm_fwd / tb_lstm_bwd launch with H > 64 may be in flight
 * on another stream while it runs (two half-resident launches would wait for each other until the 2 s timeout). */
int tb_lstm_exclusive(int B, int in_dim, int H, int backward);
size_t tb_lstm_workspace_bytes(int B, int T, int in_dim, int H, int backward);
int tb_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                void* ws, size_t ws_bytes, int* err, void* stream);
int tb_lstm_bwd(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes,
                int* err, void* stream);
/* tb_lstm_bwd with dx computed for the batch rows [dx_row_begin, B) only (rows before are left untouched): the
 * discriminator's real + fake pass reads the input gradient of the fake half alone (trainer.py:174-180).
 * (Both: for clustered layers with in_dim <= 32, a multiple of 4, dx is computed inside the BPTT kernel -- partials per
 * cluster member in the workspace, summed by a small kernel -- instead of by a GEMM over the dgates matrix.) */
int tb_lstm_bwd_rows(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                float* db_hh, int B, int T, int in_dim, int H, int accumulate, void* ws, size_t ws_bytes,
                int* err, void* stream, int dx_row_begin);

/* The weight-gradient products of tb_lstm_bwd on their own (autograd of source/utils/lstm_layer.py:44-46,83):
 * dw_ih (4H,in) = dG^T.x, dw_hh (4H,H) = dG^T.h_prev with `dgates` = the gates buffer AFTER tb_lstm_bwd.  Call
 * tb_lstm_bwd with dw_ih = dw_hh = NULL and this on another stream behind it: only dx is on the backward chain. */
size_t tb_lstm_wgrad_workspace_bytes(int in_dim, int H);
int tb_lstm_wgrad(const float* x, const float* y, const float* dgates, float* dw_ih, float* dw_hh, int B, int T,
                  int in_dim, int H, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* Inference over the OVERLAPPING stride-`stride` windows of a series (cleaning / prediction: source/cleaner.py:67-68,
 * source/predictor.py:68-69 over the windows of source/dataset.py:207-212): window b's pre-activations at step t are row
 * first_row + b * stride + t of `pseries` (series_rows x 4H) = x.W_ih^T + b_ih + b_hh computed ONCE per series row by the
 * caller (tb_linear_act_fwd with the summed bias) instead of once per (window, step) -- T / stride times fewer projection
 * flops, and the windows' encoder rows are never gathered.  y (B,T,H); pseries is only read.  Per row the same arithmetic
 * as tb_lstm_fwd on the gathered windows with the projection by GEMM: bit-identical outputs. */
int tb_lstm_fwd_windows(const float* pseries, long series_rows, long first_row, long stride, const float* w_hh, float* y,
                        int B, int T, int H, void* ws, size_t ws_bytes, int* err, void* stream);

/* tb_lstm_fwd for callers that CANNOT ALLOCATE (tight serving loops such as the cleaning pass, harnesses
 * that forbid hipMalloc, code that runs while a capture is open elsewhere).  No counterpart in the reference.  Same arguments
 * and results as tb_lstm_fwd minus the workspace: the exchange area (handshake tags + h ring) is a static
 * device array inside the library (tb_lstm_static_scratch_bytes() = 8 MiB per GPU, held until the process
 * ends); the call allocates nothing (no hipMalloc / hipFree, no events).  Shapes whose exchange area does
 * not fit are rejected with TB_ERR_WORKSPACE (4).  All calls on one GPU share that GPU's array: calls
 * that overlap in time on one GPU (other streams / threads) race -- serialise them or use tb_lstm_fwd.
 * tb_lstm_static_init() looks the array's device address up for the current device (tb_init calls it). */
int tb_lstm_static_init(void);
size_t tb_lstm_static_scratch_bytes(void);
int tb_lstm_fwd_static(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                       float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                       int* err, void* stream);

/* ---- Linear (+Tanh / +Sigmoid) -----------------------------------------------------------------
 * y[M,N] = act(x[M,K].W[N,K]^T + b[N]); act 0 none, 1 tanh, 2 sigmoid.
 * Reference: seq_layer / dim_layer / discriminator head, source/utils/lstm_layer.py:35-36,49-50,76,84-87.
 * Backward takes dz = dy*act'(y) (tb_act_bwd) and produces dx = dz.W, dW = dz^T.x, db = colsum(dz). */
int tb_linear_act_fwd(const float* x, const float* W, const float* b, float* y, int M, int K, int N, int act,
                      void* stream);
size_t tb_linear_workspace_bytes(int M, int K, int N);
int tb_linear_bwd(const float* dz, const float* x, const float* W, float* dx, float* dW, float* db, int M, int K,
                  int N, int accumulate, void* ws, size_t ws_bytes, void* stream);
int tb_act_bwd(const float* dy, const float* y, float* dz, long n, int act, void* stream);
size_t tb_colsum_workspace_bytes(int cols);
int tb_colsum(const float* a, long rows, int cols, float* out, void* ws, size_t ws_bytes, void* stream);

/* Single-output head out[m] = act(x[m,:].w + b): the discriminator's nn.Linear(hidden, 1) + Sigmoid,
 * source/utils/lstm_layer.py:76,84-87 (forward) and its autograd (backward), as streaming kernels
 * instead of a one-column GEMM.  K % 4 == 0, K <= 1024 (tb_head_supported), x / w / dx 16-byte aligned.
 * Backward: dy = gradient wrt out, out = the forward result (read when act != 0); dx[M,K], dw[K], db[1]
 * are each optional.  Workspace: tb_head_workspace_bytes(M, K). */
int tb_head_supported(int K);
size_t tb_head_workspace_bytes(long M, int K);
int tb_head_fwd(const float* x, const float* w, const float* b, float* out, long M, int K, int act, void* stream);
int tb_head_bwd(const float* dy, const float* out, const float* x, const float* w, float* dx, float* dw, float* db,
                long M, int K, int act, int accumulate, void* ws, size_t ws_bytes, void* stream);

/* ---- losses (source/loss.py) ---------------------------------------------------------------------
 * tb_loss_sums: one pass over whichever inputs are non-NULL into a device double[8]:
 *   [0] sum (Dy-0.9)^2  [1] sum (DGx-0.1)^2  [2] sum Dy  [3] sum DGx  [4] sum SmoothL1(pred-true)
 *   [5] sum (pred-true)^2.   GANLoss loss.py:34-51; l1loss/l2loss/_grad_penalty loss.py:127-141.
 *   Under data parallelism the host all-reduces this vector before finish/grad (the gradient
 *   penalty is a batch-global norm, loss.py:136-141).
 * tb_loss_finish: mode 0 dis_loss (loss.py:114-119) out=[total,real,fake,0];
 *                 mode 1 critic dis_loss (loss.py:107-112) out=[total,real,fake,gp];
 *                 mode 2 gen_loss (loss.py:93-104) out=[total,adv,l1,l2,gp].
 *   nD_*, nP, batch are GLOBAL element counts / batch size.  `running` (nullable float[8])
 *   accumulates the logging sums of loss.py:99-102,117-118; `total` (nullable float[1]) receives
 *   out[0] in separate storage (the differentiable scalar the host hands to autograd).
 * grads: `upstream` is a device float (dL/dloss, nullable = 1). */
size_t tb_loss_workspace_bytes(void);
int tb_loss_sums(const float* Dy, long nDy, const float* DGx, long nDGx, const float* tru, const float* pred,
                 long nP, double* sums, void* ws, size_t ws_bytes, void* stream);
int tb_loss_finish(const double* sums, int mode, double nD_real, double nD_fake, double nP, double batch,
                   float l1_w, float l2_w, float gp_w, float* out, float* running, float* total, void* stream);
/* tb_loss_sums + tb_loss_finish in two launches instead of three, for callers with nothing between them (no data-parallel
 * all-reduce of the sums): same arguments, same results. */
int tb_loss_sums_finish(const float* Dy, long nDy, const float* DGx, long nDGx, const float* tru, const float* pred, long nP,
                        double* sums, void* ws, size_t ws_bytes, int mode, double nD_real, double nD_fake, double nP_global,
                        double batch, float l1_w, float l2_w, float gp_w, float* out, float* running, float* total,
                        void* stream);
int tb_gan_loss_grad(const float* p, long n_local, double n_global, int target_is_real, int wgan,
                     const float* upstream, float* dp, void* stream);
/* tb_gan_loss_grad for the two halves of one buffer p = [D(real) | D(fake)] (n_each elements each, the discriminator's
 * real + fake pass) in one launch: the real half against the real label, the fake half against the fake one. */
int tb_gan_loss_grad_pair(const float* p, long n_each, double n_real_global, double n_fake_global, int wgan,
                          const float* upstream, float* dp, void* stream);
int tb_recon_loss_grad(const float* tru, const float* pred, long n_local, double n_global, double batch,
                       const double* sums, float l1_w, float l2_w, float gp_w, const float* upstream,
                       float* dpred, int accumulate, void* stream);

/* ---- optimizers over flat buffers (torch.optim defaults; schedule source/trainer.py:95-98,143-144) */
int tb_adam_step(float* p, const float* g, float* m, float* v, long n, double lr, double b1, double b2,
                 double eps, int step, void* stream);
int tb_rmsprop_step(float* p, const float* g, float* sq, long n, double lr, double alpha, double eps, int fresh,
                    void* stream);

/* Gradient-buffer housekeeping on the caller's stream (no counterpart in the reference's source: these are what
 * torch does inside optimizer.zero_grad() / autograd's accumulation, source/trainer.py:151,160-161,179,193, and
 * torch.cat).  tb_zero: bytes <- 0.  tb_grad_merge: g[i] += g2[i]; g2[i] = 0 (two backward passes of one network that
 * ran side by side into two buffers; 16-byte aligned).  tb_concat2: out = a followed by b (device-to-device copies).
 * tb_concat2_sum: out = a (na floats) followed by b + b2 (nb floats), one launch: a leaf that two losses reach gets its two
 * gradient contributions in two buffers, and their sum is formed while the joint upstream gradient is assembled. */
int tb_zero(void* p, size_t bytes, void* stream);
int tb_grad_merge(float* g, float* g2, long n, void* stream);
int tb_concat2(const void* a, size_t bytes_a, const void* b, size_t bytes_b, void* out, void* stream);
int tb_concat2_sum(const float* a, long na, const float* b, const float* b2, long nb, float* out, void* stream);

/* ---- data side ------------------------------------------------------------------------------------
 * tb_window_gather: windows of the normalised series (source/dataset.py:199-214 + utils/dataset.py:17-23):
 *   window w starts at row i = starts ? starts[w] : first + w*stride;
 *   enc[w] = series[i-To:i, :] (To,E); dec[w] = series[i:i+Tf, target_cols] (Tf,V).
 * tb_denormalize: source/dataset.py:229-242 (fp32 -> fp64 arithmetic -> fp32).
 * tb_band_update: predicts(), source/trainer.py:266-287 = source/cleaner.py:158-179 =
 *   source/predictor.py:134-155.
 *   pden (B,Tf,V) predictions; hist_in/hist_out (Tf-1,Tf,V) the predictions of the last Tf-1 windows
 *   before/after this batch (distinct buffers; nhist = valid windows in hist_in);
 *   dmin/dmax (V) float64, both or neither: if given, pden and the history hold NORMALISED predictions
 *   and tb_denormalize's arithmetic is applied as they are read (one pass less over the batch);
 *   med/sd/lower/upper (B+Tf-1,V) float64.
 * tb_adjust_scan: adjust(); variant 0 source/trainer.py:238-257, 1 source/cleaner.py:130-149,
 *   2 source/predictor.py:106-125.
 *   seed (V) = the row before reals[0]; reals/preds/lower/upper (n,V) float64; masks (n,V) int32.
 * tb_metrics: NME / NMAE / RMSE of source/metric.py:30-73 -> out[3] (+= running[3], nullable). */
/* tb_minmax: minmax_info, source/dataset.py:297-321 -- column min / max of the interpolated frame (rows, E)
 *   float64 (callers pass the first int(len*split_rate) rows; NaN skipped like pandas).  init 0 folds into what
 *   mn / mx hold (a frame streamed in row chunks); finish != 0 applies the reference's fix-ups (:311-312:
 *   min == -inf -> max, then min == max -> max + 1).  Workspace tb_minmax_workspace_bytes(E).
 * tb_normalize: normalize, source/dataset.py:216-227 -- 2*((x-min)/(max-min))-1 in float64; writes the float32
 *   series the windows are cut from (utils/dataset.py:19-20 casts every window to float32) and/or the float64
 *   frame of the reference's normalized.csv.
 * tb_take_columns: the target columns of the frame (observed / forecast, source/dataset.py:130-133). */
/* tb_interpolate: interpolate, source/dataset.py:111-115 -- forward fill, backward fill of the leading gap, 0
 *   for an empty column, IN PLACE on the float64 frame (rows, E), NaN = missing.  Optionally records the
 *   missing mask of the target columns from row row_off on (missing_label.csv, dataset.py:94-103,133):
 *   tpos (E) int32 = position of a column among the V targets or -1, missing (rows-row_off, V) int32. */
size_t tb_interpolate_workspace_bytes(long rows, int E);
int tb_interpolate(double* data, long rows, int E, const int* tpos, int V, long row_off, int* missing, void* ws,
                   size_t ws_bytes, void* stream);
size_t tb_minmax_workspace_bytes(int E);
int tb_minmax(const double* data, long rows, int E, double* mn, double* mx, int init, int finish, void* ws,
              size_t ws_bytes, void* stream);
int tb_normalize(const double* data, long rows, int E, const double* mn, const double* mx, float* out32, double* out64,
                 void* stream);
int tb_take_columns(const double* data, long rows, int E, const int* cols, int V, double* out, void* stream);
int tb_window_gather(const float* series, long series_len, int E, const long* starts, long first, long stride,
                     int nwin, int To, int Tf, const int* target_cols, int V, float* enc, float* dec,
                     void* stream);
int tb_denormalize(const float* x, long n, int V, const double* dmin, const double* dmax, float* out, void* stream);
int tb_band_update(const float* pden, int B, int Tf, int V, const float* hist_in, int nhist, float* hist_out,
                   const double* dmin, const double* dmax, double band_width, double* med, double* sd, double* lower,
                   double* upper, void* stream);
/* tb_band_update restricted to the first rows_out <= B+Tf-1 rows, with the std smear over the last Tf rows
 * (source/trainer.py:280-282) optional (it needs every row).  No counterpart in the single-device reference:
 * a data-parallel rank that is not the last of its global batch computes its own B rows unsmeared -- the
 * overhang belongs to its right neighbour.  hist_out may be NULL here. */
int tb_band_update_part(const float* pden, int B, int Tf, int V, const float* hist_in, int nhist, float* hist_out,
                        const double* dmin, const double* dmax, double band_width, double* med, double* sd,
                        double* lower, double* upper, int rows_out, int smear, void* stream);
/* tb_dimband_update: the generator's last layer (source/utils/lstm_layer.py:50, Linear(hidden, out_dim) + Tanh) and
 * tb_band_update_part in ONE kernel for the cleaning / prediction pass (source/cleaner.py:88-106): pred[w][f][:] =
 * act(hid[w][f][:] . W^T + bias) is formed per band row and sorted in registers, never written to memory.  hid (B, Tf, K)
 * fp32, W (V, K), act 0 none / 1 tanh / 2 sigmoid.  Outputs as tb_band_update_part, bit for bit; the band history is not
 * updated (the caller keeps the last Tf-1 windows' predictions).  TB_ERR_UNSUPPORTED (nothing launched) unless
 * forecast_len <= 48 and K is 32 or 64. */
int tb_dimband_update(const float* hid, int B, int Tf, int K, const float* W, const float* bias, int act, int V,
                      const float* hist_in, int nhist, const double* dmin, const double* dmax, double band_width, double* med,
                      double* sd, double* lower, double* upper, int rows_out, int smear, void* stream);
int tb_adjust_scan(int variant, const double* seed, const double* reals, const double* preds, const int* masks,
                   const double* lower, const double* upper, int n, int V, double missing_gamma,
                   double anomaly_gamma, double* target, void* stream);
int tb_metrics(const double* tru, const double* pred, const int* mask, long n, double* out, double* running,
               void* stream);

/* tb_metrics in two stages (source/metric.py:30-73): sums[5] = [sum sq err, #unmasked, sum ne, sum |ne|,
 * #unmasked non-zero] of one batch's rows -- additive over data-parallel ranks, the scores are not -- and
 * the scores of k batches from their (summed) sums[k][5]: out[k][3] (nullable), running[3] += (nullable). */
int tb_metric_sums(const double* tru, const double* pred, const int* mask, long n, double* sums, void* stream);
int tb_metric_finish(const double* sums, int k, double* out, double* running, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TIMEBAND_HIP_H */
