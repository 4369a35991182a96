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
// VGPRs for the whole sequence as MFMA A-operand fragments (W_hh never leaves the register file
// between timesteps); c stays in registers too.  The products are computed TRANSPOSED
// (gates^T = W_hh . h^T), so a lane's four accumulator registers are four consecutive hidden
// units of one batch row: every global access of the step (pre-activations, stash, y, the
// exchange) is a 16-byte access, and the write-through (sc1) exchange stores -- one fabric write
// each whatever their width -- are 4x fewer than with 4-byte stores.
// Each timestep every member needs the whole h_{t-1} tile: members write their h slice to
// y[b,t,:] with sc1 stores, raise a per-member step counter, poll the other members' counters
// (bounded spin) and read the other slices with L1-bypassing (sc1) loads -- the Guideline-16
// protocol of cdna_hip_programming.md.  In backward the same W_hh rows produce this member's
// partial of dh_{t-1} = dgates_t.W_hh; the partials are exchanged reduce-scatter style through a
// double-buffered slab.  NCL == 1 (H <= 64) needs no global exchange at all.
#include "common.h"
#include <stdlib.h>

int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream);

namespace {

constexpr unsigned kSentinel = 0xFFFFFFFFu;   // all-ones: a NaN pattern no computed h equals
constexpr int kBufFlags = 0x00020000;         // raw buffer descriptor word 3 (32-bit raw, bounds-checked)
constexpr int kSC1 = 16;                      // aux: sc1 = write-through store / L1-bypassing load

struct FwdArgs {
  float* P;            // [B][T][4H] in: pre-activations from the projection; out: activated gates
  const float* Whh;    // [4H][H]
  float* y;            // [B][T][H]
  float* cst;          // [B][T][H] cell stash (train) or nullptr
  unsigned* flags;     // [nclusters][NCL] step counters, zeroed before launch
  int* err;            // [4] sticky error words
  int B, T;
  int b_begin;         // first batch row of this launch
  int nclusters;
  int train;
  int mode;            // bit 0: y is sentinel-armed and is its own ready flag (no drain, no counter)
  unsigned long long timeout_ticks;
};

// cluster/member decode: keep the members of a cluster on blocks with equal (blockIdx % 8), i.e.
// on one XCD under the observed round-robin placement (speed only; correctness never depends on it)
__device__ __forceinline__ void cluster_decode(int NCL, int& cl, int& j) {
  const int G = gridDim.x, b = blockIdx.x;
  if ((G & 7) == 0 && ((G >> 3) % NCL) == 0) {
    const int x = b & 7, slot = b >> 3, per = (G >> 3) / NCL;
    cl = x * per + slot / NCL;
    j = slot % NCL;
  } else {
    cl = b / NCL;
    j = b % NCL;
  }
}

// Wait until every other member's counter reaches `target`.  Returns false on timeout.
__device__ __forceinline__ bool wait_members(const unsigned* flags, int NCL, int j, unsigned target,
                                             unsigned long long timeout, int* err, int* s_dead) {
  const int lane = threadIdx.x & 63;
  if (threadIdx.x < 64) {
    bool ok = true;
    if (lane < NCL && lane != j) {
      const unsigned long long t0 = wall_clock64();
      unsigned it = 0;
      while (tb_load_agent_u32(flags + lane) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++it & 1023u) == 0 && wall_clock64() - t0 > timeout) {
          ok = false;
          break;
        }
      }
    }
    if (!ok) {
      atomicExch(err, 1);
      *s_dead = 1;
    }
  }
  __syncthreads();
  return *s_dead == 0;
}

__device__ __forceinline__ float fsig(float x) { return __builtin_amdgcn_rcpf(1.f + __expf(-x)); }
__device__ __forceinline__ float ftanh(float x) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * x)); }

template <int AUX>
__device__ __forceinline__ f32x4 ldg4(__amdgpu_buffer_rsrc_t r, long float_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(float_off * 4), 0, AUX));
}
template <int AUX>
__device__ __forceinline__ void stg4(__amdgpu_buffer_rsrc_t r, long float_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), r, (int)(float_off * 4), 0, AUX);
}

template <int H, int US, int KS, int MT>
__global__ __launch_bounds__(256, 1) void lstm_fwd_kernel(FwdArgs a) {
  constexpr int NUG = US / 16;          // 16-unit groups per member
  constexpr int NCL = H / US;           // members per cluster
  constexpr int HK = H < 32 ? 32 : H;   // k extent, padded to one MFMA k-step
  constexpr int KPER = HK / KS;         // k range per wave
  constexpr int NKS = KPER / 32;        // 32-deep MFMA k-steps per wave
  constexpr int LDH = HK + 8;           // LDS row stride of the h tile (bf16 elements; 16-byte pad)
  constexpr int NACT = NUG * KS;        // waves doing MFMA work
  constexpr int RED = (KS - 1) * NUG * MT * 4 * 256;   // floats of k-split scratch
  constexpr int NV = MT * 16 * (H / 4); // 16-byte pieces of the whole h tile
  constexpr int NLD = (NV + 255) / 256;
  static_assert(NACT <= 4 && KPER % 32 == 0 && H % US == 0 && (NCL == 1 || NV % 256 == 0), "bad config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  // h_{t-1} tile as split bf16 (hi, lo): [MT*16 rows][LDH], one ds_read_b128 per MFMA B operand
  unsigned short* hiL = reinterpret_cast<unsigned short*>(smem);
  unsigned short* loL = hiL + MT * 16 * LDH;
  float* red = smem + MT * 16 * LDH;                   // [MT][NUG][KS-1][4][64][4]
  int* s_dead = reinterpret_cast<int*>(red + RED);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(NCL, cl, j);
  const bool active = wave < NACT;
  const int ug = active ? wave % NUG : 0, kp = active ? wave / NUG : 0;
  const bool fin = active && kp == 0;                  // this wave finishes the gates of its unit group
  const int uw = j * US + ug * 16 + col;               // W_hh row (per gate) this lane feeds to the MFMA
  const int u0 = j * US + ug * 16 + q * 4;             // first of the 4 hidden units this lane finishes
  const int b0 = a.b_begin + cl * 16 * MT;
  const int T = a.T;
  const long rowP = (long)T * 4 * H, rowY = (long)T * H;
  unsigned* flags = a.flags + cl * NCL;
  if (tid == 0) *s_dead = 0;
  for (int i = tid; i < MT * 16 * LDH; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;   // hi and lo, pad included
  // descriptors over this cluster's rows: loads of rows >= B return 0, stores to them are dropped
  int vrows = a.B - b0;
  if (vrows > 16 * MT) vrows = 16 * MT;
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc((void*)(a.P + (long)b0 * rowP), 0, (int)(vrows * rowP * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rY = __builtin_amdgcn_make_buffer_rsrc((void*)(a.y + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)((a.train ? a.cst : a.y) + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);

  // W_hh fragments (A operand of v_mfma_f32_16x16x32_bf16), split into bf16 hi/lo, resident for the
  // whole sequence: A[m][k] = W_hh[g*H + uw][k], k-step s of this lane covers k = kp*KPER + s*32 + q*8 .. +7
  bf16x8 whi[4][NKS], wlo[4][NKS];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int s = 0; s < NKS; ++s) {
      const int k0 = kp * KPER + s * 32 + q * 8;
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (active && k0 < H) {
        const float* wp = a.Whh + (long)(g * H + uw) * H + k0;
        v0 = *reinterpret_cast<const f32x4*>(wp);
        v1 = *reinterpret_cast<const f32x4*>(wp + 4);
      }
      tb_split8(v0, v1, whi[g][s], wlo[g][s]);
    }
  f32x4 c[MT], pre[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    c[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < 4; ++g)
      pre[mt][g] = fin ? ldg4<0>(rP, (long)(mt * 16 + col) * rowP + g * H + u0) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  __syncthreads();

  for (int t = 0; t < T; ++t) {
    // (A) next step's pre-activations: independent of the recurrence, in flight under this step
    f32x4 nxt[MT][4];
    if (t + 1 < T && fin) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) nxt[mt][g] = ldg4<0>(rP, (long)(mt * 16 + col) * rowP + (long)(t + 1) * 4 * H + g * H + u0);
    }
    // (B) (W_hh . h_{t-1}^T) on the matrix cores (h_{-1} = 0: nothing to do at t = 0)
    f32x4 acc[MT][4];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int g = 0; g < 4; ++g) acc[mt][g] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (t > 0 && active) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
          const int ho = (mt * 16 + col) * LDH + kp * KPER + s * 32 + q * 8;
          const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(hiL + ho));
          const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(loL + ho));
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[mt][g] = tb_mfma_bf16x3(whi[g][s], wlo[g][s], bh, bl, acc[mt][g]);
        }
        if (KS > 1 && kp > 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(red + ((((mt * NUG + ug) * (KS - 1) + kp - 1) * 4 + g) * 64 + lane) * 4) = acc[mt][g];
        }
      }
    }
    __syncthreads();   // all reads of hL done; k-split partials visible
    // (C) gate math, state update, h into LDS and y -- 4 consecutive units of one batch row per lane
    f32x4 st[MT][4];
    if (fin) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        f32x4 z[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          z[g] = acc[mt][g] + pre[mt][g];
          if (KS > 1 && t > 0) {
#pragma unroll
            for (int p2 = 0; p2 < KS - 1; ++p2)
              z[g] += *reinterpret_cast<const f32x4*>(red + ((((mt * NUG + ug) * (KS - 1) + p2) * 4 + g) * 64 + lane) * 4);
          }
        }
        f32x4 hn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ig = fsig(z[0][r]), fg = fsig(z[1][r]), gg = ftanh(z[2][r]), og = fsig(z[3][r]);
          const float cn = fg * c[mt][r] + ig * gg;
          c[mt][r] = cn;
          hn[r] = og * ftanh(cn);
          st[mt][0][r] = ig; st[mt][1][r] = fg; st[mt][2][r] = gg; st[mt][3][r] = og;
        }
        u32x2 sh, sl;
        tb_split4(hn, sh, sl);
        *reinterpret_cast<u32x2*>(hiL + (mt * 16 + col) * LDH + u0) = sh;
        *reinterpret_cast<u32x2*>(loL + (mt * 16 + col) * LDH + u0) = sl;
        stg4<(NCL > 1 ? kSC1 : 0)>(rY, (long)(mt * 16 + col) * rowY + (long)t * H + u0, hn);
      }
    }
    const bool exchange = NCL > 1 && t + 1 < T;
    const bool dataflag = (a.mode & 1) != 0;
    if (exchange && !dataflag) {
      // publish this member's h slice: drain ONLY the y stores (the stash stores come after the flag)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) tb_store_agent_u32(flags + j, (unsigned)(t + 1));
      asm volatile("" ::: "memory");
    }
    if (a.train && fin) {   // BPTT stash: activated gates over P in place, cell state
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int g = 0; g < 4; ++g) stg4<0>(rP, (long)(mt * 16 + col) * rowP + (long)t * 4 * H + g * H + u0, st[mt][g]);
        stg4<0>(rC, (long)(mt * 16 + col) * rowY + (long)t * H + u0, c[mt]);
      }
    }
    if (exchange) {
      u32x4 buf[NLD];
      if (!dataflag) {
        if (!wait_members(flags, NCL, j, (unsigned)(t + 1), a.timeout_ticks, a.err, s_dead)) break;
        // pull the whole tile (own slice included: one predicate-free burst) with L1-bypassing loads
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int e = tid + 256 * n;
          const int row = e & 15, c4 = (e >> 4) % (H / 4), mt = e / (16 * (H / 4));
          buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rY, (int)(((long)(mt * 16 + row) * rowY + (long)t * H + c4 * 4) * 4), 0, kSC1);
        }
      } else {
        // y was armed with the sentinel before the launch and every (b,t,u) is written exactly once,
        // so the data is its own ready flag.  One idle wave polls one piece per member first (cheap
        // on the fabric), then everyone pulls the tile and re-polls only pieces still armed.
        if (wave == 3) {
          if (lane < NCL && lane != j) {
            const int voff = (int)(((long)(vrows - 1) * rowY + (long)t * H + lane * US + US - 4) * 4);
            const unsigned long long t0 = wall_clock64();
            unsigned it = 0;
            while (true) {
              const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rY, voff, 0, kSC1);
              if (v[0] != kSentinel && v[1] != kSentinel && v[2] != kSentinel && v[3] != kSentinel) break;
              __builtin_amdgcn_s_sleep(1);
              if ((++it & 1023u) == 0 && wall_clock64() - t0 > a.timeout_ticks) {
                atomicExch(a.err, 1);
                *s_dead = 1;
                break;
              }
            }
          }
        }
        __syncthreads();
        if (*s_dead) break;
        unsigned pend = 0;
#pragma unroll
        for (int n = 0; n < NLD; ++n) {
          const int e = tid + 256 * n;
          const int c4 = (e >> 4) % (H / 4);
          if ((c4 * 4) / US != j) pend |= 1u << n;
        }
        bool first = true;
        unsigned it = 0;
        unsigned long long t0 = 0;
        while (pend) {
          asm volatile("" ::: "memory");
#pragma unroll
          for (int n = 0; n < NLD; ++n) {
            const int e = tid + 256 * n;
            const int row = e & 15, c4 = (e >> 4) % (H / 4), mt = e / (16 * (H / 4));
            if (first || ((pend >> n) & 1u))
              buf[n] = __builtin_amdgcn_raw_buffer_load_b128(rY, (int)(((long)(mt * 16 + row) * rowY + (long)t * H + c4 * 4) * 4), 0, kSC1);
          }
          first = false;
#pragma unroll
          for (int n = 0; n < NLD; ++n)
            if (buf[n][0] != kSentinel && buf[n][1] != kSentinel && buf[n][2] != kSentinel && buf[n][3] != kSentinel)
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
      }
#pragma unroll
      for (int n = 0; n < NLD; ++n) {
        const int e = tid + 256 * n;
        const int row = e & 15, c4 = (e >> 4) % (H / 4), mt = e / (16 * (H / 4));
        if ((c4 * 4) / US != j) {
          u32x2 sh, sl;
          tb_split4(__builtin_bit_cast(f32x4, buf[n]), sh, sl);
          *reinterpret_cast<u32x2*>(hiL + (mt * 16 + row) * LDH + c4 * 4) = sh;
          *reinterpret_cast<u32x2*>(loL + (mt * 16 + row) * LDH + c4 * 4) = sl;
        }
      }
    }
    __syncthreads();   // hL complete for step t+1
    if (NCL > 1 && *s_dead) break;
    if (t + 1 < T) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int g = 0; g < 4; ++g) pre[mt][g] = nxt[mt][g];
    }
  }
}

// ----------------------------------------------------------------------------------------
struct BwdArgs {
  float* G;            // [B][T][4H] in: activated gates; out: dgates (wrt pre-activations)
  const float* cst;    // [B][T][H]
  const float* dy;     // [B][T][H]
  const float* Whh;    // [4H][H]
  float* X;            // [nclusters][3][NCL consumers][NCL producers][MT*16][US] exchange slabs (NCL > 1)
  float* dbpart;       // [nclusters_total][4H] per-cluster bias-grad partials
  unsigned* flags;
  int* err;
  int B, T;
  int b_begin;
  int cl_begin;        // index of this launch's first cluster in dbpart
  int nclusters;
  int mode;
  unsigned long long timeout_ticks;
};

template <int H, int US, int MT>
__global__ __launch_bounds__(256, 1) void lstm_bwd_kernel(BwdArgs a) {
  constexpr int NCL = H / US;
  constexpr int KD = 4 * US;                       // k dimension = this member's gate rows
  constexpr int NKD = KD / 32;                     // 32-deep MFMA k-steps
  constexpr int LDG = KD + 8;                      // LDS row stride of the dgates tile (bf16 elements)
  constexpr int NTW = H >= 64 ? H / 64 : 1;        // 16-col output tiles per wave
  constexpr int NAW = H >= 64 ? 4 : H / 16;        // waves doing MFMA work
  constexpr int QPR = US / 4;                      // unit quads per row
  constexpr int NE = MT * 16 * QPR;                // (row, unit quad) items of the tile
  constexpr int NQ = (NE + 255) / 256;             // items per thread
  constexpr int LDH = H + 4;                       // dhL row stride (floats)
  static_assert((256 % QPR) == 0 && (NE % 256 == 0 || NE < 256), "bad config");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  // dgates_t of this member as split bf16 (hi, lo): [MT*16 rows][LDG]
  unsigned short* dgHi = reinterpret_cast<unsigned short*>(smem);
  unsigned short* dgLo = dgHi + MT * 16 * LDG;
  float* dhL = smem + MT * 16 * LDG;               // [MT][16][LDH] this member's partial dh_{t-1}
  float* dbL = dhL + MT * 16 * LDH;                // [256][16] bias-grad reduction
  int* s_dead = reinterpret_cast<int*>(dbL + 4096);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = lane & 15, q = lane >> 4;
  int cl, j;
  cluster_decode(NCL, cl, j);
  const int b0 = a.b_begin + cl * 16 * MT;
  const int T = a.T;
  const long rowG = (long)T * 4 * H, rowY = (long)T * H;
  unsigned* flags = a.flags + cl * NCL;
  constexpr int XSLAB = NCL * NCL * MT * 16 * US;  // floats per slab
  float* Xc = a.X + (size_t)cl * 3 * XSLAB;
  if (tid == 0) *s_dead = 0;
  int vrows = a.B - b0;
  if (vrows > 16 * MT) vrows = 16 * MT;
  const __amdgpu_buffer_rsrc_t rG = __builtin_amdgcn_make_buffer_rsrc((void*)(a.G + (long)b0 * rowG), 0, (int)(vrows * rowG * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)(a.cst + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rD = __builtin_amdgcn_make_buffer_rsrc((void*)(a.dy + (long)b0 * rowY), 0, (int)(vrows * rowY * 4), kBufFlags);
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc((void*)Xc, 0, NCL > 1 ? 3 * XSLAB * 4 : 0, kBufFlags);

  // W_hh fragments (A operand of the transposed product dh^T = W_hh^T . dgates^T), split bf16 hi/lo:
  // A[m][k] = W_hh[grow(k)][n0 + m], k = s*32 + q*8 + i, grow(k) = (k/US)*H + j*US + k%US
  bf16x8 wbh[NTW][NKD], wbl[NTW][NKD];
  const bool mactive = wave < NAW;
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int s = 0; s < NKD; ++s) {
      f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
      if (mactive) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int k = s * 32 + q * 8 + i;
          const int grow = (k / US) * H + j * US + (k % US);
          const float wv = a.Whh[(long)grow * H + (wave * NTW + nt) * 16 + col];
          if (i < 4) v0[i] = wv; else v1[i - 4] = wv;
        }
      }
      tb_split8(v0, v1, wbh[nt][s], wbl[nt][s]);
    }
  // elementwise items: e = tid + 256*n -> unit quad qd (constant per thread), row, mt
  const int qd = tid % QPR;
  const int ul0 = qd * 4, uabs0 = j * US + ul0;
  f32x4 dc[NQ], dhrec[NQ], dbacc[4];
  // stash of the current step (prefetched one step ahead): i, f, g, o, c_t, c_{t-1}, dy
  f32x4 cur[NQ][7];
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < 4; ++g) dbacc[g] = zero4;
#pragma unroll
  for (int n = 0; n < NQ; ++n) {
    dc[n] = zero4; dhrec[n] = zero4;
    const int e = tid + 256 * n;
    const int row = (e / QPR) & 15, mt = e / (16 * QPR);
    const long rr = mt * 16 + row;
    const bool ok = e < NE;
#pragma unroll
    for (int g = 0; g < 4; ++g) cur[n][g] = ok ? ldg4<0>(rG, rr * rowG + (long)(T - 1) * 4 * H + g * H + uabs0) : zero4;
    cur[n][4] = ok ? ldg4<0>(rC, rr * rowY + (long)(T - 1) * H + uabs0) : zero4;
    cur[n][5] = (ok && T > 1) ? ldg4<0>(rC, rr * rowY + (long)(T - 2) * H + uabs0) : zero4;
    cur[n][6] = ok ? ldg4<0>(rD, rr * rowY + (long)(T - 1) * H + uabs0) : zero4;
  }
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    // prefetch step t-1 (its c_t is this step's c_{t-1}, already held)
    f32x4 nx[NQ][6];
    if (t > 0) {
#pragma unroll
      for (int n = 0; n < NQ; ++n) {
        const int e = tid + 256 * n;
        const int row = (e / QPR) & 15, mt = e / (16 * QPR);
        const long rr = mt * 16 + row;
        const bool ok = e < NE;
#pragma unroll
        for (int g = 0; g < 4; ++g) nx[n][g] = ok ? ldg4<0>(rG, rr * rowG + (long)(t - 1) * 4 * H + g * H + uabs0) : zero4;
        nx[n][4] = (ok && t > 1) ? ldg4<0>(rC, rr * rowY + (long)(t - 2) * H + uabs0) : zero4;
        nx[n][5] = ok ? ldg4<0>(rD, rr * rowY + (long)(t - 1) * H + uabs0) : zero4;
      }
    }
    // (A) gate gradients for this member's units
    f32x4 dgv[NQ][4];
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float ig = cur[n][0][r], fg = cur[n][1][r], gg = cur[n][2][r], og = cur[n][3][r];
        const float dh = cur[n][6][r] + dhrec[n][r];
        const float tc = ftanh(cur[n][4][r]);
        const float dcn = dc[n][r] + dh * og * (1.f - tc * tc);
        const float dob = dh * tc * og * (1.f - og);
        const float di = dcn * gg * ig * (1.f - ig);
        const float dg = dcn * ig * (1.f - gg * gg);
        const float df = dcn * cur[n][5][r] * fg * (1.f - fg);
        dc[n][r] = dcn * fg;
        dgv[n][0][r] = di; dgv[n][1][r] = df; dgv[n][2][r] = dg; dgv[n][3][r] = dob;
      }
      if (e < NE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          dbacc[g] += dgv[n][g];
          u32x2 sh, sl;
          tb_split4(dgv[n][g], sh, sl);
          *reinterpret_cast<u32x2*>(dgHi + (mt * 16 + row) * LDG + g * US + ul0) = sh;
          *reinterpret_cast<u32x2*>(dgLo + (mt * 16 + row) * LDG + g * US + ul0) = sl;
        }
      }
    }
    const bool last = t == 0;           // dh_{-1} is never used
    const bool dataflag = (a.mode & 2) != 0;
    const int par = dataflag ? t % 3 : (t & 1);   // exchange slab of this step
    if (!last) {
      __syncthreads();
      if (NCL > 1 && *s_dead) break;   // a re-poll timed out (set before the barrier: uniform)
      // (B) partial dh_{t-1}^T = W_hh[rows of this member]^T . dgates_t^T
      if (mactive) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          f32x4 acc[NTW];
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) acc[nt] = zero4;
#pragma unroll
          for (int s = 0; s < NKD; ++s) {
            const int go = (mt * 16 + col) * LDG + s * 32 + q * 8;
            const bf16x8 bh = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgHi + go));
            const bf16x8 bl = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(dgLo + go));
#pragma unroll
            for (int nt = 0; nt < NTW; ++nt) acc[nt] = tb_mfma_bf16x3(wbh[nt][s], wbl[nt][s], bh, bl, acc[nt]);
          }
          // lane: batch row `col`, output units n0 .. n0+3
#pragma unroll
          for (int nt = 0; nt < NTW; ++nt) {
            const int n0 = (wave * NTW + nt) * 16 + q * 4;
            const int m = n0 / US;
            if (NCL == 1 || m == j) *reinterpret_cast<f32x4*>(dhL + (mt * 16 + col) * LDH + n0) = acc[nt];
            else stg4<kSC1>(rX, (long)par * XSLAB + ((long)(m * NCL + j) * MT * 16 + mt * 16 + col) * US + (n0 % US), acc[nt]);
          }
        }
      }
      if (NCL > 1 && !dataflag) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) tb_store_agent_u32(flags + j, (unsigned)(T - t));
        asm volatile("" ::: "memory");
      } else {
        __syncthreads();   // own columns of dhL visible
      }
    }
    // dgates of this step over the gate stash, in place (after the publish: not on its critical path)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
      if (e < NE) {
#pragma unroll
        for (int g = 0; g < 4; ++g) stg4<0>(rG, (long)(mt * 16 + row) * rowG + (long)t * 4 * H + g * H + uabs0, dgv[n][g]);
      }
    }
    if (last) break;
    if (NCL > 1 && !dataflag && !wait_members(flags, NCL, j, (unsigned)(T - t), a.timeout_ticks, a.err, s_dead)) break;
    if (NCL > 1 && dataflag) {
      // Data-as-flag: the slabs are armed with the sentinel (all-ones) before the launch, a slot that
      // holds a value is ready, and the consumer re-arms what it has consumed.  One wave polls one
      // piece per producer (the last one it stores), then every thread pulls its pieces and re-polls
      // only those still armed.
      if (wave == 0 && lane < NCL && lane != j) {
        const int voff = (int)(((long)par * XSLAB + ((long)(j * NCL + lane) * MT * 16 + MT * 16 - 1) * US + US - 4) * 4);
        const unsigned long long t0 = wall_clock64();
        unsigned it = 0;
        while (true) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rX, voff, 0, kSC1);
          if (v[0] != kSentinel && v[1] != kSentinel && v[2] != kSentinel && v[3] != kSentinel) break;
          __builtin_amdgcn_s_sleep(1);
          if ((++it & 1023u) == 0 && wall_clock64() - t0 > a.timeout_ticks) {
            atomicExch(a.err, 1);
            *s_dead = 1;
            break;
          }
        }
      }
      __syncthreads();
      if (*s_dead) break;
    }
    // (C) reduce-scatter: dh_{t-1} for this member's units = own partial + the other members' partials
    // (every slot is loaded, the never-written own one included: one predicate-free burst)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      const int e = tid + 256 * n;
      const int row = (e / QPR) & 15, mt = e / (16 * QPR);
      if (e < NE) {
        f32x4 s = *reinterpret_cast<const f32x4*>(dhL + (mt * 16 + row) * LDH + uabs0);
        if (NCL > 1) {
          const long base = (long)par * XSLAB + ((long)(j * NCL) * MT * 16 + mt * 16 + row) * US + ul0;
          f32x4 v[NCL];
#pragma unroll
          for (int m = 0; m < NCL; ++m) v[m] = ldg4<kSC1>(rX, base + (long)m * MT * 16 * US);
          if (dataflag) {
            unsigned pend = 0;
#pragma unroll
            for (int m = 0; m < NCL; ++m) {
              const u32x4 b = __builtin_bit_cast(u32x4, v[m]);
              if (m != j && (b[0] == kSentinel || b[1] == kSentinel || b[2] == kSentinel || b[3] == kSentinel)) pend |= 1u << m;
            }
            unsigned it = 0;
            unsigned long long t0 = 0;
            while (pend) {   // rare: a piece behind the polled one
              asm volatile("" ::: "memory");
#pragma unroll
              for (int m = 0; m < NCL; ++m)
                if ((pend >> m) & 1u) {
                  v[m] = ldg4<kSC1>(rX, base + (long)m * MT * 16 * US);
                  const u32x4 b = __builtin_bit_cast(u32x4, v[m]);
                  if (b[0] != kSentinel && b[1] != kSentinel && b[2] != kSentinel && b[3] != kSentinel) pend &= ~(1u << m);
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
          }
#pragma unroll
          for (int m = 0; m < NCL; ++m) {
            const bool other = m != j;
#pragma unroll
            for (int r = 0; r < 4; ++r) s[r] += other ? v[m][r] : 0.f;
          }
          if (dataflag) {
            // Re-arm what was consumed.  The slab of step t is written again at step t-3; its writer
            // gets there only after consuming this member's partial of step t-2, which is stored
            // after the barrier that follows the NEXT step's copy of this wait -- and that wait
            // covers these stores.  So the re-arm always lands before the next write.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const f32x4 armed = __builtin_bit_cast(f32x4, (u32x4){kSentinel, kSentinel, kSentinel, kSentinel});
#pragma unroll
            for (int m = 0; m < NCL; ++m)
              if (m != j) stg4<kSC1>(rX, base + (long)m * MT * 16 * US, armed);
          }
        }
        dhrec[n] = s;
      }
    }
    if (NCL == 1) __syncthreads();   // dhL free for the next step (NCL > 1: the waits' barriers order it)
#pragma unroll
    for (int n = 0; n < NQ; ++n) {
      cur[n][4] = cur[n][5];
      cur[n][0] = nx[n][0]; cur[n][1] = nx[n][1]; cur[n][2] = nx[n][2]; cur[n][3] = nx[n][3];
      cur[n][5] = nx[n][4]; cur[n][6] = nx[n][5];
    }
  }
  // bias grads of this cluster: reduce the threads that share a unit quad, one partial row per cluster
  __syncthreads();
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(dbL + tid * 16 + g * 4) = dbacc[g];
  __syncthreads();
  if (tid < 4 * US) {
    const int g = tid / US, ul = tid % US;
    constexpr int NTH = NE < 256 ? NE : 256;
    float s = 0.f;
    for (int th = ul / 4; th < NTH; th += QPR) s += dbL[th * 16 + g * 4 + (ul & 3)];
    a.dbpart[(size_t)(a.cl_begin + cl) * 4 * H + g * H + j * US + ul] = s;
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
    case 32: US = 32; KS = 1; break;
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
  const int HK = H < 32 ? 32 : H;
  p.lds_fwd = ((size_t)MT * 16 * (HK + 8) + (size_t)(KS - 1) * NUG * MT * 4 * 256) * 4 + 16;
  p.lds_bwd = ((size_t)MT * 16 * (4 * US + 8) + (size_t)MT * 16 * (H + 4) + 4096) * 4 + 16;
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
  FWD_CASE(32, 32, 1, 1); FWD_CASE(32, 32, 1, 2); FWD_CASE(32, 32, 1, 4);
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

// Exchange protocol, bit 0 forward / bit 1 backward: 1 = data-as-flag (sentinel-armed buffers, no
// store drain and no counter round trip; the default), 0 = drained write-through stores + per-member
// step counters (Guideline 16 R1).  TB_LSTM_MODE overrides it (A/B measurements).
int exchange_mode() {
  static int mode = -1;
  if (mode < 0) {
    const char* e = getenv("TB_LSTM_MODE");
    mode = e ? atoi(e) : 3;
  }
  return mode;
}

// the kernels address one cluster's rows through 32-bit buffer descriptors
bool tile_fits_descriptor(const Plan& p, int T, int H) {
  return (long long)p.rows_per_cluster * T * 4 * H * 4 < 0x7FFFFFFFll;
}

// workspace carve: [flags | bias sum | dbpart | X | gemm split-k slabs]
struct Carve {
  unsigned* flags; size_t flags_bytes;
  float* bsum;
  float* dbpart;
  float* X; size_t x_bytes;
  float* gemm_ws; size_t gemm_ws_bytes;
};

size_t carve(const Plan& p, int B, int H, int in_dim, bool bwd, char* base, size_t total, Carve* c) {
  const int ncl_total = (B + p.rows_per_cluster - 1) / p.rows_per_cluster;
  size_t off = 0;
  const size_t flags_bytes = tb_align_up((size_t)ncl_total * p.NCL * sizeof(unsigned), 256);
  const size_t bsum_bytes = tb_align_up((size_t)4 * H * sizeof(float), 256);
  const size_t dbpart_bytes = bwd ? tb_align_up((size_t)ncl_total * 4 * H * sizeof(float), 256) : 0;
  const int ncl_launch = ncl_total < p.clusters_per_launch ? ncl_total : p.clusters_per_launch;
  const size_t x_bytes = (bwd && p.NCL > 1)
                             ? tb_align_up((size_t)ncl_launch * 3 * p.NCL * p.rows_per_cluster * H * sizeof(float), 256)
                             : 0;
  const int widest = in_dim > H ? in_dim : H;
  const size_t gemm_bytes = bwd ? tb_align_up((size_t)64 * 4 * H * widest * sizeof(float), 256) : 0;
  if (c) {
    c->flags = reinterpret_cast<unsigned*>(base + off); c->flags_bytes = flags_bytes;
  }
  off += flags_bytes;
  if (c) c->bsum = reinterpret_cast<float*>(base + off);
  off += bsum_bytes;
  if (c) c->dbpart = reinterpret_cast<float*>(base + off);
  off += dbpart_bytes;
  if (c) { c->X = reinterpret_cast<float*>(base + off); c->x_bytes = x_bytes; }
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
  if (!tile_fits_descriptor(p, T, H)) return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_fwd: T*H too large (%d x %d)", T, H);
  const int mode = exchange_mode();
  TB_HIP(hipMemsetAsync(c.flags, 0, c.flags_bytes, s));
  if (p.NCL > 1 && (mode & 1)) TB_HIP(hipMemsetAsync(y, 0xFF, (size_t)B * T * H * sizeof(float), s));
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
    a.flags = c.flags + (size_t)c0 * p.NCL;
    a.err = err;
    a.B = B; a.T = T;
    a.b_begin = c0 * p.rows_per_cluster;
    a.nclusters = ncl;
    a.train = train;
    a.mode = mode;
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
  if (!tile_fits_descriptor(p, T, H)) return tb_set_error(TB_ERR_UNSUPPORTED, "lstm_bwd: T*H too large (%d x %d)", T, H);
  TB_HIP(hipMemsetAsync(c.flags, 0, c.flags_bytes, s));
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
    a.flags = c.flags + (size_t)c0 * p.NCL;
    a.err = err;
    a.B = B; a.T = T;
    a.b_begin = c0 * p.rows_per_cluster;
    a.cl_begin = c0;
    a.mode = exchange_mode();
    if (p.NCL > 1 && (a.mode & 2)) TB_HIP(hipMemsetAsync(c.X, 0xFF, c.x_bytes, s));   // arm the slabs
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
