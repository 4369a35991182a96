// K2 / K3: persistent LSTM recurrent forward and fused backward-through-time for gfx950.
//
// Replaces aten::lstm / its autograd (cuDNN or oneDNN under torch) as called by the reference
// at source/utils/lstm_layer.py:31-33,44-46 (generator lstm0/1/2) and :75,83 (discriminator):
// torch.nn.LSTM semantics, gate order i,f,g,o, two biases, batch_first, zero initial state.
//
// Decomposition (one layer):
//   forward : P = x.W_ih^T + b_ih + b_hh        (K1 GEMM, all timesteps at once)
//             persistent recurrent kernel over t (this file): a = P_t + h_{t-1}.W_hh^T,
//             fused sigmoid/tanh, c/h update; activated gates overwrite P in place (BPTT stash)
//   backward: persistent BPTT kernel over t (this file): dgates_t overwrite the stash in place,
//             dh_{t-1} = dgates_t.W_hh; bias grads accumulated in registers
//             then K1 GEMMs: dx = dG.W_ih, dW_ih = dG^T.x, dW_hh = dG^T.h_prev
//
// Persistent kernel layout.  The batch is cut into tiles of 16*MT rows; a CLUSTER of NCL = H/US
// workgroups (one per CU, all co-resident: grid <= #CUs) owns one tile.  Member j owns hidden
// units [j*US, (j+1)*US): in forward the 4*US rows of W_hh that produce its gates, held in
// VGPRs for the whole sequence as MFMA B-operand fragments (W_hh never leaves the register file
// between timesteps); c stays in registers too.  Each timestep every member needs the whole
// h_{t-1} tile: members write their h slice to y[b,t,:] with write-through (sc1) stores, raise a
// per-member step counter, poll the other members' counters (bounded spin) and read the other
// slices with L1-bypassing (sc1) loads -- the Guideline-16 protocol of cdna_hip_programming.md.
// In backward the same W_hh rows are the B operand of dh_{t-1} = dgates_t.W_hh; the per-member
// partial sums over gate rows are exchanged reduce-scatter style through a double-buffered slab.
// NCL == 1 (H <= 64) needs no global exchange at all.
#include "common.h"

int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream);

namespace {

struct FwdArgs {
  float* P;            // [B][T][4H] in: pre-activations from the projection; out: activated gates
  const float* Whh;    // [4H][H]
  float* y;            // [B][T][H]
  float* cst;          // [B][T][H] cell stash (train) or nullptr
  int* err;            // [4] sticky error words
  int B, T;
  int b_begin;         // first batch row of this launch
  int nclusters;
  int train;
  unsigned long long timeout_ticks;
};

__device__ __forceinline__ int hl_off(int row, int k) {
  // LDS image of a [16][K] tile laid out so that one ds_read_b128 per lane yields the MFMA A
  // operand of 4 consecutive k-steps: float offset of element (row, k).
  return (((k >> 4) * 64 + ((k >> 2) & 3) * 16 + row) << 2) + (k & 3);
}

// cluster/member decode: keep the members of a cluster on blocks with equal (blockIdx % 8), i.e.
// on one XCD under the observed round-robin placement (speed only; correctness never depends on it)
__device__ __forceinline__ void cluster_decode(int ncl, int NCL, int& cl, int& j) {
  const int G = gridDim.x, b = blockIdx.x;
  if ((G & 7) == 0 && ((G >> 3) % NCL) == 0) {
    const int x = b & 7, slot = b >> 3, per = (G >> 3) / NCL;
    cl = x * per + slot / NCL;
    j = slot % NCL;
  } else {
    cl = b / NCL;
    j = b % NCL;
  }
  (void)ncl;
}

constexpr unsigned kSentinel = 0xFFFFFFFFu;   // all-ones: a NaN pattern no finite h equals

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

// element r0+rr of a 4-row accumulator, r0 = kp*RPW, without dynamic register indexing
template <int KS>
__device__ __forceinline__ float pick_row(const f32x4& v, int kp, int rr) {
  if (KS == 1) return v[rr];
  if (KS == 2) return kp ? v[2 + rr] : v[rr];
  return kp == 0 ? v[0] : (kp == 1 ? v[1] : (kp == 2 ? v[2] : v[3]));
}

template <int H, int US, int KS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_fwd_kernel(FwdArgs a) {
  constexpr int NUG = US / 16;          // 16-unit groups per member
  constexpr int NCL = H / US;           // members per cluster
  constexpr int KPER = H / KS;          // k range per wave
  constexpr int KG = KPER / 16;         // 16-wide k groups per wave
  constexpr int NACT = NUG * KS;        // waves doing work
  constexpr int RPW = 4 / KS;           // accumulator rows each k-split partner finishes
  constexpr int RED = KS > 1 ? KS * NUG * MT * 4 * 256 : 0;   // floats of k-split scratch
  static_assert(NACT <= 4 && KPER % 16 == 0 && H % US == 0, "bad config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* hL = smem;                                    // [MT][H/16][64][4]
  float* red = smem + MT * H * 16;                     // [MT][NUG][KS][4][64][4]
  int* s_dead = reinterpret_cast<int*>(smem + MT * H * 16 + RED);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(a.nclusters, NCL, cl, j);
  const bool active = wave < NACT;
  const int ug = active ? wave % NUG : 0, kp = active ? wave / NUG : 0;
  const int u = j * US + ug * 16 + col;                // hidden unit of this lane
  const int b0 = a.b_begin + cl * 16 * MT;
  const int T = a.T;
  const long rowP = (long)T * 4 * H, rowY = (long)T * H;
  if (tid == 0) *s_dead = 0;

  // W_hh fragments: B[k][n] = W_hh[g*H + u][k]; MFMA step (kg,i) uses k = kp*KPER + kg*16 + q*4 + i
  float w[4][KG * 4];
  if (active) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int kg = 0; kg < KG; ++kg) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(a.Whh + (long)(g * H + u) * H + kp * KPER + kg * 16 + q * 4);
        w[g][kg * 4 + 0] = v[0]; w[g][kg * 4 + 1] = v[1]; w[g][kg * 4 + 2] = v[2]; w[g][kg * 4 + 3] = v[3];
      }
  }
  // this wave finishes accumulator rows [kp*RPW, kp*RPW+RPW) of every 4-row group
  float c[MT][RPW], pre[MT][4][RPW];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      c[mt][rr] = 0.f;
      const int b = b0 + mt * 16 + q * 4 + kp * RPW + rr;
      const bool ok = active && b < a.B;
      const float* p = a.P + (long)(ok ? b : 0) * rowP + u;
#pragma unroll
      for (int g = 0; g < 4; ++g) pre[mt][g][rr] = ok ? p[g * H] : 0.f;
    }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // (A) next step's pre-activations: independent of the recurrence, in flight under this step
    float nxt[MT][4][RPW];
    if (t + 1 < T) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int b = b0 + mt * 16 + q * 4 + kp * RPW + rr;
          const bool ok = active && b < a.B;
          const float* p = a.P + (long)(ok ? b : 0) * rowP + (long)(t + 1) * 4 * H + u;
#pragma unroll
          for (int g = 0; g < 4; ++g) nxt[mt][g][rr] = ok ? p[g * H] : 0.f;
        }
    }
    // (B) h_{t-1}.W_hh^T on the matrix cores (h_{-1} = 0: nothing to do at t = 0)
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[mt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t > 0 && active) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int kg = 0; kg < KG; ++kg) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(hL + ((mt * (H / 16) + kp * KG + kg) * 64 + lane) * 4);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
              acc[mt][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], w[g][kg * 4 + i], acc[mt][g], 0, 0, 0);
        }
        if (KS > 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(red + ((((mt * NUG + ug) * KS + kp) * 4 + g) * 64 + lane) * 4) = acc[mt][g];
        }
      }
    }
    __syncthreads();   // all reads of hL done; k-split partials visible
    // (C) gate math for this wave's rows, state update, h into LDS and y
    float st[MT][5][RPW];
    if (active) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          float z[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            z[g] = pick_row<KS>(acc[mt][g], kp, rr) + pre[mt][g][rr];
            if (KS > 1 && t > 0) {
#pragma unroll
              for (int p2 = 0; p2 < KS; ++p2)
                if (p2 != kp) z[g] += red[((((mt * NUG + ug) * KS + p2) * 4 + g) * 64 + lane) * 4 + kp * RPW + rr];
            }
          }
          const int row = q * 4 + kp * RPW + rr;
          const int b = b0 + mt * 16 + row;
          const float ig = fsig(z[0]), fg = fsig(z[1]), gg = ftanh(z[2]), og = fsig(z[3]);
          const float cn = fg * c[mt][rr] + ig * gg;
          c[mt][rr] = cn;
          const float hn = og * ftanh(cn);
          st[mt][0][rr] = ig; st[mt][1][rr] = fg; st[mt][2][rr] = gg; st[mt][3][rr] = og; st[mt][4][rr] = cn;
          hL[mt * H * 16 + hl_off(row, u)] = hn;
          if (b < a.B) {
            float* yp = a.y + (long)b * rowY + (long)t * H + u;
            if (NCL > 1) tb_store_agent(yp, hn);
            else *yp = hn;
          }
        }
      }
    }
    const bool exchange = NCL > 1 && t + 1 < T;
    if (exchange) {
      // Pull the other members' slices of h_t.  y was armed with the sentinel (all-ones, a NaN no
      // finite h can equal) before the launch and every (b,t,u) is written exactly once, so the data
      // is its own ready flag: poll with L1-bypassing 16-byte loads until no dword is the sentinel.
      int vrows = a.B - b0;
      if (vrows > 16 * MT) vrows = 16 * MT;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)(a.y + (long)b0 * rowY), 0, (int)((long)vrows * rowY * 4), 0x00020000);
      constexpr int NV = MT * 16 * (H / 4);       // float4 pieces of the whole h tile
      constexpr int NLD = (NV + 255) / 256;
      u32x4 buf[NLD];
      unsigned pend = 0;
#pragma unroll
      for (int n = 0; n < NLD; ++n) {
        const int e = tid + 256 * n;
        const int c4 = (e >> 4) % (H / 4);
        if (e < NV && (c4 * 4) / US != j) pend |= 1u << n;
      }
      const unsigned want = pend;
      unsigned it = 0;
      unsigned long long t0 = 0;
      while (pend) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int e = tid + 256 * n;
          const int row = e & 15, c4 = (e >> 4) % (H / 4), mt = e / (16 * (H / 4));
          if ((pend >> n) & 1u)
            buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((long)(mt * 16 + row) * rowY + (long)t * H + c4 * 4) * 4), 0, 16);
        }
#pragma unroll
        for (int n = 0; n < NLD; ++n)
          if (((pend >> n) & 1u) && buf[n][0] != kSentinel && buf[n][1] != kSentinel && buf[n][2] != kSentinel &&
              buf[n][3] != kSentinel)
            pend &= ~(1u << n);
        if (pend && (++it & 255u) == 0) {
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > a.timeout_ticks) {
            atomicExch(a.err, 1);
            *s_dead = 1;
            break;
          }
        }
      }
#pragma unroll
      for (int n = 0; n < NLD; ++n) {
        const int e = tid + 256 * n;
        const int row = e & 15, c4 = (e >> 4) % (H / 4), mt = e / (16 * (H / 4));
        if ((want >> n) & 1u) *reinterpret_cast<u32x4*>(hL + mt * H * 16 + hl_off(row, c4 * 4)) = buf[n];
      }
    }
    if (a.train && active) {   // BPTT stash (off the critical path): activated gates over P in place, cell state
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int b = b0 + mt * 16 + q * 4 + kp * RPW + rr;
          if (b < a.B) {
            float* p = a.P + (long)b * rowP + (long)t * 4 * H + u;
            p[0] = st[mt][0][rr]; p[H] = st[mt][1][rr]; p[2 * H] = st[mt][2][rr]; p[3 * H] = st[mt][3][rr];
            a.cst[(long)b * rowY + (long)t * H + u] = st[mt][4][rr];
          }
        }
    }
    __syncthreads();   // hL complete for step t+1
    if (NCL > 1 && *s_dead) break;   // a member timed out (uniform: read after the barrier)
    if (t + 1 < T) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
          for (int rr = 0; rr < RPW; ++rr) pre[mt][g][rr] = nxt[mt][g][rr];
    }
  }
}

// ----------------------------------------------------------------------------------------
struct BwdArgs {
  float* G;            // [B][T][4H] in: activated gates; out: dgates (wrt pre-activations)
  const float* cst;    // [B][T][H]
  const float* dy;     // [B][T][H]
  const float* Whh;    // [4H][H]
  unsigned long long* X;   // [nclusters][2][MT*16][H][NCL] (value, tag) exchange slabs (NCL > 1)
  float* dbpart;       // [nclusters_total][4H] per-cluster bias-grad partials
  int* err;
  int B, T;
  int b_begin;
  int cl_begin;        // index of this launch's first cluster in dbpart
  int nclusters;
  unsigned long long timeout_ticks;
};

template <int H, int US, int MT>
__global__ __launch_bounds__(256, 1) void lstm_bwd_kernel(BwdArgs a) {
  constexpr int NCL = H / US;
  constexpr int KD = 4 * US;                       // k dimension = this member's gate rows
  constexpr int KGD = KD / 16;
  constexpr int NTW = H >= 64 ? H / 64 : 1;        // 16-col output tiles per wave
  constexpr int NAW = H >= 64 ? 4 : H / 16;        // waves doing MFMA work
  constexpr int NI = MT * 16 * US / 256;           // (row,unit) items per thread
  constexpr int LDH = H + 4;                       // dhL row stride (floats)
  static_assert(NI >= 1 && (256 % US) == 0 && (NCL == 1 || NCL % 2 == 0) && NI * (NCL > 1 ? NCL / 2 : 1) <= 32, "bad config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* dgL = smem;                               // [MT][KD/16][64][4]
  float* dhL = smem + MT * KD * 16;                // [MT][16][LDH] this member's partial dh_{t-1}
  float* dbL = dhL + MT * 16 * LDH;                // [256][4] bias-grad reduction
  int* s_dead = reinterpret_cast<int*>(dbL + 1024);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(a.nclusters, NCL, cl, j);
  const int b0 = a.b_begin + cl * 16 * MT;
  const int T = a.T;
  const long rowG = (long)T * 4 * H, rowY = (long)T * H;
  // exchange slab of this cluster: [parity][MT*16 rows][H cols][NCL members] (member fastest, so the
  // reduce-scatter reads NCL contiguous floats per (row, unit))
  unsigned long long* Xc = a.X + (size_t)cl * 2 * MT * 16 * H * NCL;
  if (tid == 0) *s_dead = 0;

  // W_hh fragments: B[k][n] = W_hh[grow(k)][n], k = kg*16 + q*4 + i, grow(k) = (k/US)*H + j*US + k%US
  float wb[NTW][KGD * 4];
  const bool mactive = wave < NAW;
  if (mactive) {
#pragma unroll
    for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
      for (int s = 0; s < KGD * 4; ++s) {
        const int k = (s >> 2) * 16 + q * 4 + (s & 3);
        const int grow = (k / US) * H + j * US + (k % US);
        wb[nt][s] = a.Whh[(long)grow * H + (wave * NTW + nt) * 16 + col];
      }
  }
  // elementwise items: e = tid + 256*n -> ul = e % US (constant per thread), row, mt
  const int ul = tid % US;
  const int uabs = j * US + ul;
  float dc[NI], dhrec[NI], dbacc[4] = {0.f, 0.f, 0.f, 0.f};
  // stash of the current step (prefetched one step ahead): i, f, g, o, c_t, c_{t-1}, dy
  float cur[NI][7];
#pragma unroll
  for (int n = 0; n < NI; ++n) {
    dc[n] = 0.f; dhrec[n] = 0.f;
    const int e = tid + 256 * n;
    const int row = (e / US) & 15, mt = e / (16 * US);
    const int b = b0 + mt * 16 + row;
    const bool ok = b < a.B;
    const long bb = ok ? b : 0;
    const float* gp = a.G + bb * rowG + (long)(T - 1) * 4 * H + uabs;
    cur[n][0] = ok ? gp[0] : 0.f; cur[n][1] = ok ? gp[H] : 0.f;
    cur[n][2] = ok ? gp[2 * H] : 0.f; cur[n][3] = ok ? gp[3 * H] : 0.f;
    cur[n][4] = ok ? a.cst[bb * rowY + (long)(T - 1) * H + uabs] : 0.f;
    cur[n][5] = (ok && T > 1) ? a.cst[bb * rowY + (long)(T - 2) * H + uabs] : 0.f;
    cur[n][6] = ok ? a.dy[bb * rowY + (long)(T - 1) * H + uabs] : 0.f;
  }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    // prefetch step t-1 (its c_t is this step's c_{t-1}, already held)
    float nx[NI][6];
    if (t > 0) {
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int e = tid + 256 * n;
        const int row = (e / US) & 15, mt = e / (16 * US);
        const int b = b0 + mt * 16 + row;
        const bool ok = b < a.B;
        const long bb = ok ? b : 0;
        const float* gp = a.G + bb * rowG + (long)(t - 1) * 4 * H + uabs;
        nx[n][0] = ok ? gp[0] : 0.f; nx[n][1] = ok ? gp[H] : 0.f;
        nx[n][2] = ok ? gp[2 * H] : 0.f; nx[n][3] = ok ? gp[3 * H] : 0.f;
        nx[n][4] = (ok && t > 1) ? a.cst[bb * rowY + (long)(t - 2) * H + uabs] : 0.f;
        nx[n][5] = ok ? a.dy[bb * rowY + (long)(t - 1) * H + uabs] : 0.f;
      }
    }
    // (A) gate gradients for this member's units
    float dgv[NI][4];
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / US) & 15, mt = e / (16 * US);
      const float ig = cur[n][0], fg = cur[n][1], gg = cur[n][2], og = cur[n][3];
      const float dh = cur[n][6] + dhrec[n];
      const float tc = ftanh(cur[n][4]);
      const float dcn = dc[n] + dh * og * (1.f - tc * tc);
      const float dob = dh * tc * og * (1.f - og);
      const float di = dcn * gg * ig * (1.f - ig);
      const float dg = dcn * ig * (1.f - gg * gg);
      const float df = dcn * cur[n][5] * fg * (1.f - fg);
      dc[n] = dcn * fg;
      dgv[n][0] = di; dgv[n][1] = df; dgv[n][2] = dg; dgv[n][3] = dob;
      dbacc[0] += di; dbacc[1] += df; dbacc[2] += dg; dbacc[3] += dob;
      float* dl = dgL + mt * KD * 16;
      dl[hl_off(row, ul)] = di;
      dl[hl_off(row, US + ul)] = df;
      dl[hl_off(row, 2 * US + ul)] = dg;
      dl[hl_off(row, 3 * US + ul)] = dob;
    }
    const bool last = t == 0;           // dh_{-1} is never used
    const int par = t & 1;
    const unsigned tag = (unsigned)(T - t);
    if (!last) {
      __syncthreads();
      // (B) partial dh_{t-1} = dgates_t . W_hh[rows of this member]
      if (mactive) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 acc[NTW];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int kg = 0; kg < KGD; ++kg) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(dgL + ((mt * KGD + kg) * 64 + lane) * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int nt = 0; nt < NTW; ++nt)
                acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], wb[nt][kg * 4 + i], acc[nt], 0, 0, 0);
          }
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const int ncol = (wave * NTW + nt) * 16 + col;
            const bool own = (ncol / US) == j;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = q * 4 + r;
              if (NCL == 1 || own) dhL[(mt * 16 + row) * LDH + ncol] = acc[nt][r];
              else   // (value, step tag) in one 8-byte write-through store: the pair is its own ready flag
                tb_store_agent_u64(Xc + (((size_t)par * MT * 16 + mt * 16 + row) * H + ncol) * NCL + j,
                                   ((unsigned long long)tag << 32) | __float_as_uint(acc[nt][r]));
            }
          }
        }
      }
      __syncthreads();   // own columns of dhL visible
    }
    // dgates of this step over the gate stash, in place (after the publish: not on its critical path)
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / US) & 15, mt = e / (16 * US);
      const int b = b0 + mt * 16 + row;
      if (b < a.B) {
        float* gp = a.G + (long)b * rowG + (long)t * 4 * H + uabs;
        gp[0] = dgv[n][0]; gp[H] = dgv[n][1]; gp[2 * H] = dgv[n][2]; gp[3 * H] = dgv[n][3];
      }
    }
    if (last) break;
    // (C) reduce-scatter: dh_{t-1} for this member's units = sum over members of their partials.
    // Every slab slot is an 8-byte (value, tag) pair written by one store, so it is its own ready
    // flag (no store drain, no separate flag round trip): poll with L1-bypassing 16-byte loads (two
    // pairs each) until every foreign tag equals this step's.  The slab is zeroed before the launch
    // and tags start at 1.  Two parities: a slot is rewritten two steps later, and its writer gets
    // there only after consuming this member's partial of the step in between, which this member
    // produces after it has finished reading here.
    {
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
          (void*)Xc, 0, (int)((size_t)2 * MT * 16 * H * NCL * 8), 0x00020000);
      constexpr int NQ = NCL > 1 ? NCL / 2 : 1;
      u32x4 buf[NI][NQ];
      unsigned pend = NCL > 1 ? (1u << (NI * NQ)) - 1u : 0u;
      unsigned it = 0;
      unsigned long long t0 = 0;
      while (pend) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int n = 0; n < NI; ++n) {
          const int e = tid + 256 * n;
          const int row = (e / US) & 15, mt = e / (16 * US);
          const int base = (int)(((((size_t)par * MT * 16 + mt * 16 + row) * H + uabs) * NCL) * 8);
#pragma unroll
          for (int m2 = 0; m2 < NQ; ++m2)
            if ((pend >> (n * NQ + m2)) & 1u) buf[n][m2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, base + m2 * 16, 0, 16);
        }
#pragma unroll
        for (int n = 0; n < NI; ++n)
#pragma unroll
          for (int m2 = 0; m2 < NQ; ++m2) {
            const bool ready = (m2 * 2 == j || buf[n][m2][1] == tag) && (m2 * 2 + 1 == j || buf[n][m2][3] == tag);
            if (ready) pend &= ~(1u << (n * NQ + m2));
          }
        if (pend && (++it & 255u) == 0) {
          const unsigned long long now = wall_clock64();
          if (t0 == 0) t0 = now;
          else if (now - t0 > a.timeout_ticks) {
            atomicExch(a.err, 1);
            *s_dead = 1;
            break;
          }
        }
      }
#pragma unroll
      for (int n = 0; n < NI; ++n) {
        const int e = tid + 256 * n;
        const int row = (e / US) & 15, mt = e / (16 * US);
        float s = dhL[(mt * 16 + row) * LDH + uabs];
        if (NCL > 1) {
#pragma unroll
          for (int m2 = 0; m2 < NQ; ++m2) {
            if (m2 * 2 != j) s += __uint_as_float(buf[n][m2][0]);
            if (m2 * 2 + 1 != j) s += __uint_as_float(buf[n][m2][2]);
          }
        }
        dhrec[n] = s;
      }
    }
    if (NCL > 1) {
      // a timed-out member must stop the whole workgroup: make the verdict uniform
      __syncthreads();
      if (*s_dead) break;
    }
#pragma unroll
    for (int n = 0; n < NI; ++n) {
      cur[n][4] = cur[n][5];
      cur[n][0] = nx[n][0]; cur[n][1] = nx[n][1]; cur[n][2] = nx[n][2]; cur[n][3] = nx[n][3];
      cur[n][5] = nx[n][4]; cur[n][6] = nx[n][5];
    }
  }
  // bias grads of this cluster: reduce the threads that share a unit, one partial row per cluster
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) dbL[tid * 4 + g] = dbacc[g];
  __syncthreads();
  if (tid < US) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = tid; k < 256; k += US)
#pragma unroll
      for (int g = 0; g < 4; ++g) s[g] += dbL[k * 4 + g];
    float* out = a.dbpart + (size_t)(a.cl_begin + cl) * 4 * H;
#pragma unroll
    for (int g = 0; g < 4; ++g) out[g * H + j * US + tid] = s[g];
  }
}

__global__ void db_reduce_kernel(const float* __restrict__ part, int nparts, int n, float* db_ih, float* db_hh) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int p = 0; p < nparts; ++p) s += part[(size_t)p * n + i];
  if (db_ih) db_ih[i] = s;
  if (db_hh) db_hh[i] = s;
}

__global__ void bias_sum_kernel(const float* b1, const float* b2, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = b1[i] + b2[i];
}

// ---- launch planning ---------------------------------------------------------------------
struct Plan {
  int H, US, KS, MT;
  int NCL;
  int rows_per_cluster;
  int clusters_per_launch;
  size_t lds_fwd, lds_bwd;
};

bool make_plan(int H, int B, Plan& p) {
  int US, KS;
  switch (H) {
    case 16: US = 16; KS = 1; break;
    case 32: US = 32; KS = 2; break;
    case 64: US = 64; KS = 1; break;
    case 128: US = 32; KS = 2; break;
    case 256: US = 32; KS = 2; break;
    case 512: US = 16; KS = 4; break;
    default: return false;
  }
  p.H = H; p.US = US; p.KS = KS; p.NCL = H / US;
  const int cus = tb_num_cus();
  const int max_clusters = cus / p.NCL > 0 ? cus / p.NCL : 1;
  int MT = 1;
  const int mt_cap = (H >= 512) ? 1 : ((H >= 256 || H == 64) ? 2 : 4);
  while (MT < mt_cap && (B + 16 * MT - 1) / (16 * MT) > max_clusters) MT *= 2;
  p.MT = MT;
  p.rows_per_cluster = 16 * MT;
  p.clusters_per_launch = max_clusters;
  const int NUG = US / 16;
  p.lds_fwd = ((size_t)MT * H * 16 + (size_t)(KS > 1 ? KS : 0) * NUG * MT * 4 * 256) * 4 + 16;
  p.lds_bwd = ((size_t)MT * 4 * US * 16 + (size_t)MT * 16 * (H + 4) + 1024) * 4 + 16;
  return true;
}

template <int H, int US, int KS, int MT>
int launch_fwd_t(const FwdArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_fwd_kernel<H, US, KS, MT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_fwd_kernel<H, US, KS, MT>), dim3(grid), dim3(256), lds, s, a);
  TB_LAUNCH_CHECK("lstm_fwd_kernel");
  return TB_OK;
}

template <int H, int US, int MT>
int launch_bwd_t(const BwdArgs& a, int grid, size_t lds, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&lstm_bwd_kernel<H, US, MT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL((lstm_bwd_kernel<H, US, MT>), dim3(grid), dim3(256), lds, s, a);
  TB_LAUNCH_CHECK("lstm_bwd_kernel");
  return TB_OK;
}

#define FWD_CASE(h, us, ks, mt) \
  if (p.H == h && p.MT == mt) return launch_fwd_t<h, us, ks, mt>(a, grid, p.lds_fwd, s)
#define BWD_CASE(h, us, mt) \
  if (p.H == h && p.MT == mt) return launch_bwd_t<h, us, mt>(a, grid, p.lds_bwd, s)

int launch_fwd(const Plan& p, const FwdArgs& a, int grid, hipStream_t s) {
  FWD_CASE(16, 16, 1, 1); FWD_CASE(16, 16, 1, 2); FWD_CASE(16, 16, 1, 4);
  FWD_CASE(32, 32, 2, 1); FWD_CASE(32, 32, 2, 2); FWD_CASE(32, 32, 2, 4);
  FWD_CASE(64, 64, 1, 1); FWD_CASE(64, 64, 1, 2);
  FWD_CASE(128, 32, 2, 1); FWD_CASE(128, 32, 2, 2); FWD_CASE(128, 32, 2, 4);
  FWD_CASE(256, 32, 2, 1); FWD_CASE(256, 32, 2, 2);
  FWD_CASE(512, 16, 4, 1);
  return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: no kernel for H=%d MT=%d", p.H, p.MT);
}

int launch_bwd(const Plan& p, const BwdArgs& a, int grid, hipStream_t s) {
  BWD_CASE(16, 16, 1); BWD_CASE(16, 16, 2); BWD_CASE(16, 16, 4);
  BWD_CASE(32, 32, 1); BWD_CASE(32, 32, 2); BWD_CASE(32, 32, 4);
  BWD_CASE(64, 64, 1); BWD_CASE(64, 64, 2);
  BWD_CASE(128, 32, 1); BWD_CASE(128, 32, 2); BWD_CASE(128, 32, 4);
  BWD_CASE(256, 32, 1); BWD_CASE(256, 32, 2);
  BWD_CASE(512, 16, 1);
  return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: no kernel for H=%d MT=%d", p.H, p.MT);
}

constexpr unsigned long long kTimeoutTicks = 200000000ull;   // 2 s of the 100 MHz realtime counter

// workspace carve: [bias sum | dbpart | X | gemm split-k slabs]
struct Carve {
  float* bsum;
  float* dbpart;
  unsigned long long* X; size_t x_bytes;
  float* gemm_ws; size_t gemm_ws_bytes;
};

size_t carve(const Plan& p, int B, int H, int in_dim, bool bwd, char* base, size_t total, Carve* c) {
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  size_t off = 0;
  const size_t bsum_bytes = tb_align_up((size_t)4 * H * sizeof(float), 256);
  const size_t dbpart_bytes = bwd ? tb_align_up((size_t)ncl_total * 4 * H * sizeof(float), 256) : 0;
  const int ncl_launch = ncl_total < p.clusters_per_launch ? ncl_total : p.clusters_per_launch;
  const size_t x_bytes = (bwd && p.NCL > 1)
                             ? tb_align_up((size_t)ncl_launch * 2 * p.NCL * p.rows_per_cluster * H * sizeof(unsigned long long), 256)
                             : 0;
  const int widest = in_dim > H ? in_dim : H;
  const size_t gemm_bytes = bwd ? tb_align_up((size_t)64 * 4 * H * widest * sizeof(float), 256) : 0;
  if (c) c->bsum = reinterpret_cast<float*>(base + off);
  off += bsum_bytes;
  if (c) c->dbpart = reinterpret_cast<float*>(base + off);
  off += dbpart_bytes;
  if (c) { c->X = reinterpret_cast<unsigned long long*>(base + off); c->x_bytes = x_bytes; }
  off += x_bytes;
  if (c) {
    c->gemm_ws = reinterpret_cast<float*>(base + off);
    c->gemm_ws_bytes = total > off ? total - off : 0;
  }
  off += gemm_bytes;
  return off;
}

}  // namespace

extern "C" {

size_t tb_lstm_workspace_bytes(int B, int T, int in_dim, int H, int backward) {
  (void)T;
  Plan p;
  if (!make_plan(H, B, p)) return 0;
  return carve(p, B, H, in_dim, backward != 0, nullptr, 0, nullptr);
}

int tb_lstm_supported(int H) {
  Plan p;
  return make_plan(H, 16, p) ? 1 : 0;
}

int tb_lstm_fwd(const float* x, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                float* y, float* gates, float* cstash, int B, int T, int in_dim, int H, int train,
                void* ws, size_t ws_bytes, int* err, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0 || T <= 0) return TB_OK;
  if (!x || !w_ih || !w_hh || !y || !gates || !ws || !err) return tb_set_error(TB_ERR_ARG, "lstm_fwd: null pointer");
  if (train && !cstash) return tb_set_error(TB_ERR_ARG, "lstm_fwd: train needs a cell stash");
  Plan p;
  if (!make_plan(H, B, p))
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: hidden size %d not supported (16,32,64,128,256,512)", H);
  Carve c;
  const size_t need = carve(p, B, H, in_dim, false, static_cast<char*>(ws), ws_bytes, &c);
  if (need > ws_bytes) return tb_set_error(TB_ERR_WORKSPACE, "lstm_fwd: workspace %zu < %zu", ws_bytes, need);
  // arm y with the sentinel the cluster members poll on (see lstm_fwd_kernel)
  if (p.NCL > 1) TB_HIP(hipMemsetAsync(y, 0xFF, (size_t)B * T * H * sizeof(float), s));
  // P = x.W_ih^T + (b_ih + b_hh), written straight into the gate stash
  const float* bias = nullptr;
  if (b_ih && b_hh) {
    hipLaunchKernelGGL(bias_sum_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, s, b_ih, b_hh, c.bsum, 4 * H);
    TB_LAUNCH_CHECK("bias_sum_kernel");
    bias = c.bsum;
  }
  int rc = tb_gemm_launch(x, in_dim, 0, w_ih, in_dim, 0, gates, 4L * H, bias, B * T, 4 * H, in_dim, 0, 0, 1,
                          nullptr, 0, s);
  if (rc) return rc;
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  // algorithmic bytes of one layer forward: read x, write h, write the 5h stash
  tb_prof_begin(TB_K_LSTM_FWD, s, 4.0 * B * T * (train ? (in_dim + 6.0 * H) : (in_dim + 1.0 * H)));
  for (int c0 = 0; c0 < ncl_total; c0 += p.clusters_per_launch) {
    const int ncl = (ncl_total - c0) < p.clusters_per_launch ? (ncl_total - c0) : p.clusters_per_launch;
    FwdArgs a;
    a.P = gates; a.Whh = w_hh; a.y = y; a.cst = cstash;
    a.err = err;
    a.B = B; a.T = T;
    a.b_begin = c0 * p.rows_per_cluster;
    a.nclusters = ncl;
    a.train = train;
    a.timeout_ticks = kTimeoutTicks;
    rc = launch_fwd(p, a, ncl * p.NCL, s);
    if (rc) return rc;
  }
  tb_prof_end(TB_K_LSTM_FWD, s);
  return TB_OK;
}

int tb_lstm_bwd(const float* dy, const float* x, const float* w_ih, const float* w_hh, const float* y,
                float* gates, const float* cstash, float* dx, float* dw_ih, float* dw_hh, float* db_ih,
                float* db_hh, int B, int T, int in_dim, int H, void* ws, size_t ws_bytes, int* err,
                void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (B <= 0 || T <= 0) return TB_OK;
  if (!dy || !x || !w_ih || !w_hh || !y || !gates || !cstash || !ws || !err)
    return tb_set_error(TB_ERR_ARG, "lstm_bwd: null pointer");
  Plan p;
  if (!make_plan(H, B, p))
    return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: hidden size %d not supported (16,32,64,128,256,512)", H);
  Carve c;
  const size_t need = carve(p, B, H, in_dim, true, static_cast<char*>(ws), ws_bytes, &c);
  if (need > ws_bytes) return tb_set_error(TB_ERR_WORKSPACE, "lstm_bwd: workspace %zu < %zu", ws_bytes, need);
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  int rc;
  // algorithmic bytes of one layer backward: read 5h stash + dy, read x, write dx
  tb_prof_begin(TB_K_LSTM_BWD, s, 4.0 * B * T * (2.0 * in_dim + 6.0 * H));
  for (int c0 = 0; c0 < ncl_total; c0 += p.clusters_per_launch) {
    const int ncl = (ncl_total - c0) < p.clusters_per_launch ? (ncl_total - c0) : p.clusters_per_launch;
    BwdArgs a;
    a.G = gates; a.cst = cstash; a.dy = dy; a.Whh = w_hh;
    a.X = c.X;
    a.dbpart = c.dbpart;
    a.err = err;
    a.B = B; a.T = T;
    a.b_begin = c0 * p.rows_per_cluster;
    a.cl_begin = c0;
    if (p.NCL > 1) TB_HIP(hipMemsetAsync(c.X, 0, c.x_bytes, s));   // tags start below 1
    a.nclusters = ncl;
    a.timeout_ticks = kTimeoutTicks;
    rc = launch_bwd(p, a, ncl * p.NCL, s);
    if (rc) return rc;
  }
  tb_prof_end(TB_K_LSTM_BWD, s);
  if (db_ih || db_hh) {
    hipLaunchKernelGGL(db_reduce_kernel, dim3((4 * H + 255) / 256), dim3(256), 0, s, c.dbpart, ncl_total, 4 * H,
                       db_ih, db_hh);
    TB_LAUNCH_CHECK("db_reduce_kernel");
  }
  const int M = B * T;
  if (dx) {   // dx = dG . W_ih            [M,4H].[4H,in]
    rc = tb_gemm_launch(gates, 4L * H, 0, w_ih, in_dim, 1, dx, in_dim, nullptr, M, in_dim, 4 * H, 0, 0, 1,
                        c.gemm_ws, c.gemm_ws_bytes, s);
    if (rc) return rc;
  }
  if (dw_ih) {  // dW_ih = dG^T . x        [4H,M].[M,in]
    rc = tb_gemm_launch(gates, 4L * H, 1, x, in_dim, 1, dw_ih, in_dim, nullptr, 4 * H, in_dim, M, 0, 0, 1,
                        c.gemm_ws, c.gemm_ws_bytes, s);
    if (rc) return rc;
  }
  if (dw_hh) {  // dW_hh = dG^T . h_prev,  h_prev[b,t] = y[b,t-1], 0 at t = 0
    rc = tb_gemm_launch(gates, 4L * H, 1, y, H, 1, dw_hh, H, nullptr, 4 * H, H, M, 0, 1, T, c.gemm_ws,
                        c.gemm_ws_bytes, s);
    if (rc) return rc;
  }
  return TB_OK;
}

}  // extern "C"
