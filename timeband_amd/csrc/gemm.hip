// K1: fp32-grade GEMM on the gfx950 matrix cores with split-precision bf16 operands ("bf16x3").
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
// Operands stay fp32 in HBM.  Each fp32 tile is split in registers into bf16 hi + bf16 lo
// (common.h) on its way into LDS, and every 16x16x32 product is three v_mfma_f32_16x16x32_bf16
// (hi.hi + hi.lo + lo.hi, fp32 accumulate): ~2^-17 relative operand error, i.e. fp32-grade results,
// at 16/3 of the fp32 MFMA rate -- which moves these GEMMs (K <= 128 projections, K = B.T weight
// gradients over a 268 MB dgates matrix) from MFMA-bound to HBM-bound.
// Block tile BM x BN x 32 with 4 waves laid out WGM x WGN, each wave a (BM/WGM) x (BN/WGN) patch of
// 16x16 MFMA tiles.  Tiles keep their global orientation in LDS (no transposing stores, 8-byte
// ds_writes): a k-contiguous tile is [rows][32+8] and one ds_read_b128 per lane is an MFMA operand;
// an m-contiguous tile is [32][rows+8] and the operand is two ds_read_b64_tr_b16 (the gfx950
// hardware transpose read).  Next tile's global loads fly under the MFMAs.
// Split-K (gridDim.z) writes partial slabs that splitk_reduce sums in a fixed order (bitwise
// reproducible; no float atomics).
#include "common.h"

namespace {

constexpr int BK = 32;
typedef short s16x4 __attribute__((ext_vector_type(4)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  const float* bias;
  const float* bias2;   // optional second bias vector: the epilogue adds (bias + bias2)
  int M, N, K;
  long lda, ldb, ldc;
  int act;
  int kchunk;
  int splits;
  float* slab;
  int b_shift, b_period;
  int vecA, vecB;
  int bytesA, bytesB;    // extents of the operands for the bounds-checked buffer loads
  int xcd_map;           // gemm_big_kernel: XCD-aware item numbering (TB_GEMM_XMAP=0: off)
};

template <int LAY, int ROWS>
struct Tile {
  static constexpr int LD = LAY == 0 ? BK + 8 : ROWS + 8;            // LDS row stride (bf16 elements)
  static constexpr int ELEMS = LAY == 0 ? ROWS * LD : BK * LD;       // one image (hi or lo)
  static constexpr int NCHUNK = ROWS * BK / 4;                       // 16-byte fp32 chunks of one tile
  static constexpr int NCH = (NCHUNK + 255) / 256;                   // chunks per thread
  static_assert(NCHUNK % 256 == 0 || NCHUNK < 256, "chunk count must tile the block");

  // chunk c of this thread -> (row, k) of its first element
  static __device__ __forceinline__ void where(int c, int& r, int& k) {
    const int e = threadIdx.x + 256 * c;
    if (LAY == 0) { r = e >> 3; k = (e & 7) * 4; }
    else { k = e / (ROWS / 4); r = (e % (ROWS / 4)) * 4; }
  }

  // Branch-free tile load through a bounds-checked buffer descriptor: an element outside the matrix
  // gets an offset past the descriptor's end and reads as 0.  (With per-lane branches around the loads
  // hipcc waits for each load before it issues the next one: no prefetch at all.)
  template <bool VEC>
  static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, long ld, int r0, int rlim, int k0, int klim,
                                              int shift, int period, f32x4 (&v)[NCH]) {
    constexpr int kOob = 0x7FFFFFF0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      r += r0; k += k0;
      bool ok = r < rlim && k < klim && !(NCHUNK < 256 && (int)threadIdx.x >= NCHUNK);
      if (LAY == 1 && shift) {   // B row k is source row k - shift, zero for the first `shift` rows of each period
        ok = ok && (k % period) >= shift;
        k -= shift;
      }
      const long e0 = LAY == 0 ? (long)r * ld + k : (long)k * ld + r;
      if (VEC) {
        v[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (int)(e0 * 4) : kOob, 0, 0));
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const bool oki = ok && (LAY == 0 ? k + i < klim : r + i < rlim);
          v[c][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, oki ? (int)((e0 + i) * 4) : kOob, 0, 0));
        }
      }
    }
  }

  // split into bf16 hi / lo and store in the global orientation (8-byte LDS writes)
  static __device__ __forceinline__ void store(unsigned short* hi, unsigned short* lo, const f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      if (NCHUNK < 256 && threadIdx.x >= NCHUNK) continue;
      u32x2 sh, sl;
      tb_split4(v[c], sh, sl);
      const int off = LAY == 0 ? r * LD + k : k * LD + r;
      *reinterpret_cast<u32x2*>(hi + off) = sh;
      *reinterpret_cast<u32x2*>(lo + off) = sl;
    }
  }

  // MFMA operand (8 consecutive k of row `row + col`, k = q*8 ..) from one image
  static __device__ __forceinline__ bf16x8 frag(const unsigned short* img, int row, int col, int q) {
    if (LAY == 0) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + (row + col) * LD + q * 8));
    // [k][row] image: ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of a 4-row block;
    // lane 4*qq+p of the group supplies the address of block row qq, columns 4p..4p+3
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const unsigned short* p0 = img + (q * 8 + (col >> 2)) * LD + row + 4 * (col & 3);
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * LD));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    return __builtin_bit_cast(bf16x8, (u32x4){ua[0], ua[1], ub[0], ub[1]});
  }
};

// FAST: both operands and the output are 16-byte addressable (pointers and leading dimensions multiples
// of 4 floats, N a multiple of 4): one b128 load per chunk, one b128 store per epilogue piece, no
// per-lane branches anywhere in the loop.  Otherwise dword loads / scalar tail stores.
template <int LAY_A, int LAY_B, int BM, int BN, int WGM, int WGN, bool FAST, int PF = 1>
__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(GemmArgs a) {
  using TA = Tile<LAY_A, BM>;
  using TB = Tile<LAY_B, BN>;
  constexpr int TM = BM / WGM / 16, TN = BN / WGN / 16;
  static_assert(WGM * WGN == 4 && TM >= 1 && TN >= 1, "bad wave layout");
  __shared__ __attribute__((aligned(16))) unsigned short Ah[TA::ELEMS], Al[TA::ELEMS];
  __shared__ __attribute__((aligned(16))) unsigned short Bh[TB::ELEMS], Bl[TB::ELEMS];
  constexpr int CW = TN * 16, PLD = CW + 4;             // epilogue patch: 16 rows x CW columns per wave
  static_assert(64 % (CW / 4) == 0, "a lane must keep its epilogue columns over the patch rows");
  __shared__ __attribute__((aligned(16))) float patch[4][16 * PLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int col = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;
  // Persistent over work items (output tile x k-split), n fastest: concurrent blocks then write whole
  // rows of C between them (a row of C is a DRAM/TLB page at N = 1024; m-fastest order touched 128 pages
  // per tile with 512 B each) and share the A tile in L2; B is the small operand and stays L2-resident.  The first k-tile of the NEXT item is loaded under the last MFMAs and the epilogue of
  // the current one: small-K products (one or two k-tiles per item) pipeline across items.
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.N + BN - 1) / BN;
  const int total = mtiles * ntiles * a.splits;
  int w = blockIdx.x;
  if (w >= total) return;
  auto decode = [&](int item, int& m0, int& n0, int& z, int& kbeg, int& kend) {
    z = item / (mtiles * ntiles);
    const int r = item % (mtiles * ntiles);
    // n fastest, rotated by the m tile: blocks b, b+8, ... share an XCD, and with a power-of-two number
    // of n tiles they would all write the same columns of C -- the same few L2 / HBM channels
    m0 = (r / ntiles) * BM;
    n0 = ((r + r / ntiles) % ntiles) * BN;
    kbeg = z * a.kchunk;
    kend = min(a.K, kbeg + a.kchunk);
  };
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, a.bytesB, 0x00020000);
  if constexpr (PF == 2) {
    // Operand tiles fetched TWO k-tiles ahead.  With one tile in flight per workgroup (two per CU) the loop ran at
    // the latency of a first-touch HBM read: 2.3 us per k-tile against 0.36 us of MFMA issue -- neither the
    // products, nor the bf16 split, nor the stores bounded it (a build without MFMAs took 0.188 ms where the full
    // one takes 0.208).  The stream of (item, k-tile) pairs of this block is flattened: `ld` runs two tiles
    // ahead of `cp` across item boundaries, the two register sets are refilled right after their tile has gone
    // to LDS, and the A fragments are read per row block to make room for the second set.
    struct Pos { int w, m0, n0, z, k, kend; bool valid; };
    auto first = [&](Pos& p, int w_) {
      p.w = w_;
      p.valid = w_ < total;
      p.m0 = p.n0 = p.z = p.k = p.kend = 0;
      if (p.valid) decode(w_, p.m0, p.n0, p.z, p.k, p.kend);
    };
    auto advance = [&](Pos& p) {
      p.k += BK;
      if (p.k >= p.kend) first(p, p.w + (int)gridDim.x);
    };
    Pos ld, cp;
    first(ld, w);
    cp = ld;
    f32x4 ra0[TA::NCH], rb0[TB::NCH], ra1[TA::NCH], rb1[TB::NCH];
    auto issue = [&](f32x4 (&ra)[TA::NCH], f32x4 (&rb)[TB::NCH]) {
      if (ld.valid) {
        TA::template load<FAST>(rA, a.lda, ld.m0, a.M, ld.k, ld.kend, 0, 1, ra);
        TB::template load<FAST>(rB, a.ldb, ld.n0, a.N, ld.k, ld.kend, a.b_shift, a.b_period, rb);
        advance(ld);
      }
    };
    issue(ra0, rb0);
    issue(ra1, rb1);
    f32x4 acc[TM][TN];
    f32x4 bv;
    auto new_item = [&]() {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (a.bias && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb2_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias2, 0, (a.bias2 && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const int nb = cp.n0 + wn * CW + (lane % (CW / 4)) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_, (nb + r) * 4, 0, 0)) +
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2_, (nb + r) * 4, 0, 0));
    };
    new_item();
    auto half = [&](f32x4 (&ra)[TA::NCH], f32x4 (&rb)[TB::NCH]) -> bool {
      TA::store(Ah, Al, ra);
      TB::store(Bh, Bl, rb);
      __syncthreads();
      issue(ra, rb);          // the tile two ahead, into the registers that have just been emptied
      bf16x8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = TB::frag(Bh, (wn * TN + j) * 16, col, q);
        bl[j] = TB::frag(Bl, (wn * TN + j) * 16, col, q);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bf16x8 ah = TA::frag(Ah, (wm * TM + i) * 16, col, q);
        const bf16x8 al = TA::frag(Al, (wm * TM + i) * 16, col, q);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
      }
      __syncthreads();   // all fragment reads done before the next tile overwrites the images
      if (cp.k + BK >= cp.kend) {
        // epilogue of the finished item (as in the one-tile-ahead form below)
        float* out = a.splits > 1 ? a.slab + (size_t)cp.z * a.M * a.N : a.C;
        const long ldo = a.splits > 1 ? a.N : a.ldc;
        float* pw = patch[wave];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
          for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(pw + col * PLD + j * 16 + q * 4) = acc[i][j];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < TN; ++k) {
            const int idx = k * 64 + lane, row = idx / (CW / 4), c4 = idx % (CW / 4);
            f32x4 v = *reinterpret_cast<const f32x4*>(pw + row * PLD + c4 * 4);
            const int m = cp.m0 + (wm * TM + i) * 16 + row;
            const int n = cp.n0 + wn * CW + c4 * 4;
            if (m >= a.M || n >= a.N) continue;
            if (a.splits == 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                v[r] += bv[r];
                if (a.act == 1) v[r] = tb_tanh(v[r]);
                else if (a.act == 2) v[r] = tb_sigmoid(v[r]);
              }
            }
            float* po = out + (long)m * ldo + n;
            if (FAST) *reinterpret_cast<f32x4*>(po) = v;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (n + r < a.N) po[r] = v[r];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        advance(cp);
        if (!cp.valid) return true;
        new_item();
        return false;
      }
      advance(cp);
      return false;
    };
    while (true) {
      if (half(ra0, rb0)) break;
      if (half(ra1, rb1)) break;
    }
    return;
  }
  int m0, n0, z, kbeg, kend;
  decode(w, m0, n0, z, kbeg, kend);
  f32x4 ra[TA::NCH], rb[TB::NCH];
  TA::template load<FAST>(rA, a.lda, m0, a.M, kbeg, kend, 0, 1, ra);
  TB::template load<FAST>(rB, a.ldb, n0, a.N, kbeg, kend, a.b_shift, a.b_period, rb);
  while (true) {
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // bias of this lane's 4 epilogue columns (bounds-checked dword loads: 0 past N or without a bias),
    // fetched before the k loop: a load inside the epilogue would drain every store issued before it
    f32x4 bv;
    {
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (a.bias && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb2_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias2, 0, (a.bias2 && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const int nb = n0 + wn * CW + (lane % (CW / 4)) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_, (nb + r) * 4, 0, 0)) +
                __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2_, (nb + r) * 4, 0, 0));
      // wait HERE (the operand tile loaded before it is needed now anyway): left to the compiler, the wait
      // lands at the first use in the epilogue as vmcnt(0) and drains the stores issued before it
      __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    }
    const int wnext = w + gridDim.x;
    int m0n = 0, n0n = 0, zn = 0, kbegn = 0, kendn = 0;
    if (wnext < total) decode(wnext, m0n, n0n, zn, kbegn, kendn);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
      TA::store(Ah, Al, ra);
      TB::store(Bh, Bl, rb);
      __syncthreads();
      if (k0 + BK < kend) {   // next k-tile's global loads fly under this tile's MFMAs
        TA::template load<FAST>(rA, a.lda, m0, a.M, k0 + BK, kend, 0, 1, ra);
        TB::template load<FAST>(rB, a.ldb, n0, a.N, k0 + BK, kend, a.b_shift, a.b_period, rb);
      } else if (wnext < total) {
        TA::template load<FAST>(rA, a.lda, m0n, a.M, kbegn, kendn, 0, 1, ra);
        TB::template load<FAST>(rB, a.ldb, n0n, a.N, kbegn, kendn, a.b_shift, a.b_period, rb);
      }
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        ah[i] = TA::frag(Ah, (wm * TM + i) * 16, col, q);
        al[i] = TA::frag(Al, (wm * TM + i) * 16, col, q);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = TB::frag(Bh, (wn * TN + j) * 16, col, q);
        bl[j] = TB::frag(Bl, (wn * TN + j) * 16, col, q);
      }
      // three passes over the independent accumulators: no MFMA waits on the one before it.  The products
      // are computed transposed (C^T = B^T.A^T: the B fragment is the MFMA's row operand), so a lane's four
      // accumulator registers are four consecutive n of one output row: 16-byte epilogue stores.
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al[i], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah[i], acc[i][j], 0, 0, 0);
      __syncthreads();   // all fragment reads done before the next tile overwrites the images
    }

    // Epilogue.  A lane holds 4 consecutive n of row m = col, so a direct store would send the 64 lanes of
    // one instruction to 16 different rows (64 separate 16-byte writes).  Each wave turns its 16 x CW
    // patches through a private LDS buffer instead, so that 16 consecutive lanes write one contiguous
    // 256-byte piece of a row (LDS operations of one wave execute in order: no barrier needed).
    float* out = a.splits > 1 ? a.slab + (size_t)z * a.M * a.N : a.C;
    const long ldo = a.splits > 1 ? a.N : a.ldc;
    float* pw = patch[wave];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(pw + col * PLD + j * 16 + q * 4) = acc[i][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < TN; ++k) {
        const int idx = k * 64 + lane, row = idx / (CW / 4), c4 = idx % (CW / 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(pw + row * PLD + c4 * 4);
        const int m = m0 + (wm * TM + i) * 16 + row;
        const int n = n0 + wn * CW + c4 * 4;
        if (m >= a.M || n >= a.N) continue;
#ifdef TB_GEMM_NOSTORE
        if (a.act != 77) continue;   // timing experiment: no global stores
#endif
        if (a.splits == 1) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            v[r] += bv[r];
            if (a.act == 1) v[r] = tb_tanh(v[r]);
            else if (a.act == 2) v[r] = tb_sigmoid(v[r]);
          }
        }
        float* po = out + (long)m * ldo + n;
        if (FAST) *reinterpret_cast<f32x4*>(po) = v;
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (n + r < a.N) po[r] = v[r];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (wnext >= total) break;
    w = wnext; m0 = m0n; n0 = n0n; z = zn; kbeg = kbegn; kend = kendn;
  }
}

// ---- the large-K products (weight gradients: K = rows x steps; input gradients: K = 4H) -----------------------------
// The 128 x 128 kernel above spends most of its time off the matrix pipe: per MFMA it converts and stores 43 operand
// elements (3 vector instructions each for the bf16 hi / lo split, ds_write_b64 at ~85 B/clk) and crosses two
// barriers per 32-deep k-tile with one LDS image.  This variant is for products whose k loop is long enough to
// amortise a bigger tile: 256 x 128 per workgroup of 8 waves (4 x 2, each wave the same 64 x 64 patch as above, two
// waves per SIMD), 32 elements per MFMA; LDS images double-buffered -- the next tile is stored while the current one is
// read, ONE barrier per k-tile --; operand tiles fetched two k-tiles ahead into two register sets.  One workgroup per
// CU (123 KB of LDS).  Measured (tools/micro/wgrad_micro.py, hidden 256, 512 rows x 128 steps): dW_hh 199 -> 157 us,
// dW_ih 97 -> 76, dx 96 -> 83; the iteration 6.22 -> 6.03 ms.  Tried and NOT kept: fixed wave roles (waves 0-3 MFMA on
// 128 x 64 patches, waves 4-7 load / split / store: 239 us -- the four loader waves cannot keep two tiles in flight
// AND convert them in the time the MFMA waves need); a per-workgroup rotation of the k loop against HBM channel
// camping (no change).
template <int LAY, int ROWS, int NT>
struct TileN {
  static constexpr int LD = LAY == 0 ? BK + 8 : ROWS + 8;
  static constexpr int ELEMS = LAY == 0 ? ROWS * LD : BK * LD;
  static constexpr int NCHUNK = ROWS * BK / 4;
  static constexpr int NCH = NCHUNK / NT;
  static_assert(NCHUNK % NT == 0, "chunk count must tile the block");
  static __device__ __forceinline__ void where(int c, int& r, int& k) {
    const int e = threadIdx.x + NT * c;
    if (LAY == 0) { r = e >> 3; k = (e & 7) * 4; }
    else { k = e / (ROWS / 4); r = (e % (ROWS / 4)) * 4; }
  }
  static __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, long ld, int r0, int rlim, int k0, int klim,
                                              int shift, int period, f32x4 (&v)[NCH]) {
    constexpr int kOob = 0x7FFFFFF0;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      r += r0; k += k0;
      bool ok = r < rlim && k < klim;
      if (LAY == 1 && shift) {
        ok = ok && (k % period) >= shift;
        k -= shift;
      }
      const long e0 = LAY == 0 ? (long)r * ld + k : (long)k * ld + r;
      v[c] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (int)(e0 * 4) : kOob, 0, 0));
    }
  }
  static __device__ __forceinline__ void store(unsigned short* hi, unsigned short* lo, const f32x4 (&v)[NCH]) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int r, k;
      where(c, r, k);
      u32x2 sh, sl;
      tb_split4(v[c], sh, sl);
      const int off = LAY == 0 ? r * LD + k : k * LD + r;
      *reinterpret_cast<u32x2*>(hi + off) = sh;
      *reinterpret_cast<u32x2*>(lo + off) = sl;
    }
  }
  static __device__ __forceinline__ bf16x8 frag(const unsigned short* img, int row, int col, int q) {
    if (LAY == 0) return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(img + (row + col) * LD + q * 8));
    typedef s16x4 __attribute__((address_space(3))) * lds_s16x4;
    const unsigned short* p0 = img + (q * 8 + (col >> 2)) * LD + row + 4 * (col & 3);
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(p0 + 4 * LD));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    return __builtin_bit_cast(bf16x8, (u32x4){ua[0], ua[1], ub[0], ub[1]});
  }
};

constexpr int kBigBM = 256, kBigBN = 128, kBigNT = 512;
template <int LAY_A, int LAY_B>
constexpr size_t big_lds_bytes() {
  return (size_t)2 * 2 * (TileN<LAY_A, kBigBM, kBigNT>::ELEMS + TileN<LAY_B, kBigBN, kBigNT>::ELEMS) * sizeof(unsigned short) +
         (size_t)8 * 16 * (64 + 4) * sizeof(float);
}

template <int LAY_A, int LAY_B>
__global__ __launch_bounds__(512, 1) void gemm_big_kernel(GemmArgs a) {
  constexpr int BM = kBigBM, BN = kBigBN, NT = kBigNT, WGN = 2, TM = 4, TN = 4;
  using TA = TileN<LAY_A, BM, NT>;
  using TB = TileN<LAY_B, BN, NT>;
  constexpr int CW = TN * 16, PLD = CW + 4;
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  // [image][Ah | Al | Bh | Bl], then the epilogue patches
  constexpr int BUF = 2 * (TA::ELEMS + TB::ELEMS);
  float* patch_all = reinterpret_cast<float*>(lds + 2 * BUF);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int col = lane & 15, q = lane >> 4;
  const int wm = wave / WGN, wn = wave % WGN;
  const int mtiles = (a.M + BM - 1) / BM, ntiles = (a.N + BN - 1) / BN;
  const int total = mtiles * ntiles * a.splits;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)a.A, 0, a.bytesA, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)a.B, 0, a.bytesB, 0x00020000);
  // Work items.  The tiles of ONE k-split read the same rows of A and B: they go to workgroups of one XCD (blocks with
  // equal blockIdx % 8 under the round-robin placement), so that the XCD's L2 serves the re-reads (2-4 %).
  const int T_ = mtiles * ntiles;
  const bool xmap = (gridDim.x & 7) == 0 && a.splits > 1 && a.xcd_map;
  const int vtotal = xmap ? T_ * ((a.splits + 7) / 8) * 8 : total;
  for (int w = blockIdx.x; w < vtotal; w += gridDim.x) {
    int z, r;
    if (xmap) {
      const int slot = w >> 3;
      z = (w & 7) + 8 * (slot / T_);
      r = slot % T_;
      if (z >= a.splits) continue;
    } else {
      z = w / T_;
      r = w % T_;
    }
    const int m0 = (r / ntiles) * BM, n0 = ((r + r / ntiles) % ntiles) * BN;
    const int kbeg = z * a.kchunk, kend = min(a.K, kbeg + a.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    f32x4 ra0[TA::NCH], rb0[TB::NCH], ra1[TA::NCH], rb1[TB::NCH];
    auto issue = [&](int kt, f32x4 (&ra)[TA::NCH], f32x4 (&rb)[TB::NCH]) {
      if (kt < nk) {
        TA::load(rA, a.lda, m0, a.M, kbeg + kt * BK, kend, 0, 1, ra);
        TB::load(rB, a.ldb, n0, a.N, kbeg + kt * BK, kend, a.b_shift, a.b_period, rb);
      }
    };
    auto images = [&](int b, unsigned short*& Ah, unsigned short*& Al, unsigned short*& Bh, unsigned short*& Bl) {
      Ah = lds + b * BUF; Al = Ah + TA::ELEMS; Bh = Al + TA::ELEMS; Bl = Bh + TB::ELEMS;
    };
    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bv;
    {
      const __amdgpu_buffer_rsrc_t rb_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias, 0, (a.bias && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const __amdgpu_buffer_rsrc_t rb2_ = __builtin_amdgcn_make_buffer_rsrc((void*)a.bias2, 0, (a.bias2 && a.splits == 1) ? a.N * 4 : 0, 0x00020000);
      const int nb = n0 + wn * CW + (lane % (CW / 4)) * 4;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        bv[r4] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb_, (nb + r4) * 4, 0, 0)) +
                 __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rb2_, (nb + r4) * 4, 0, 0));
    }
    issue(0, ra0, rb0);
    issue(1, ra1, rb1);
    {
      unsigned short *Ah, *Al, *Bh, *Bl;
      images(0, Ah, Al, Bh, Bl);
      TA::store(Ah, Al, ra0);
      TB::store(Bh, Bl, rb0);
    }
    issue(2, ra0, rb0);
    __syncthreads();
    // one k-tile: MFMAs on image `kt & 1`; the tile after it (already in `ra`/`rb`) goes to the other image meanwhile,
    // and the registers it leaves are refilled with the tile three ahead
    auto half = [&](int kt, f32x4 (&ra)[TA::NCH], f32x4 (&rb)[TB::NCH]) {
      unsigned short *Ah, *Al, *Bh, *Bl, *nAh, *nAl, *nBh, *nBl;
      images(kt & 1, Ah, Al, Bh, Bl);
      images((kt + 1) & 1, nAh, nAl, nBh, nBl);
      bf16x8 bh[TN], bl[TN];
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        bh[j] = TB::frag(Bh, (wn * TN + j) * 16, col, q);
        bl[j] = TB::frag(Bl, (wn * TN + j) * 16, col, q);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const bf16x8 ah = TA::frag(Ah, (wm * TM + i) * 16, col, q);
        const bf16x8 al = TA::frag(Al, (wm * TM + i) * 16, col, q);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bl[j], ah, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], al, acc[i][j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bh[j], ah, acc[i][j], 0, 0, 0);
        if (i == 0 && kt + 1 < nk) TA::store(nAh, nAl, ra);
        if (i == 1 && kt + 1 < nk) TB::store(nBh, nBl, rb);
        if (i == 2) issue(kt + 3, ra, rb);
      }
      __syncthreads();
    };
    for (int kt = 0; kt < nk; kt += 2) {
      half(kt, ra1, rb1);
      if (kt + 1 < nk) half(kt + 1, ra0, rb0);
    }
    // epilogue: each wave turns its 16 x 64 patches through a private LDS buffer (rows of 256 contiguous bytes)
    float* out = a.splits > 1 ? a.slab + (size_t)z * a.M * a.N : a.C;
    const long ldo = a.splits > 1 ? a.N : a.ldc;
    float* pw = patch_all + wave * 16 * PLD;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN; ++j) *reinterpret_cast<f32x4*>(pw + col * PLD + j * 16 + q * 4) = acc[i][j];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < TN; ++k) {
        const int idx = k * 64 + lane, row = idx / (CW / 4), c4 = idx % (CW / 4);
        f32x4 v = *reinterpret_cast<const f32x4*>(pw + row * PLD + c4 * 4);
        const int m = m0 + (wm * TM + i) * 16 + row;
        const int n = n0 + wn * CW + c4 * 4;
        if (m >= a.M || n >= a.N) continue;
        if (a.splits == 1) {
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) {
            v[r4] += bv[r4];
            if (a.act == 1) v[r4] = tb_tanh(v[r4]);
            else if (a.act == 2) v[r4] = tb_sigmoid(v[r4]);
          }
        }
        *reinterpret_cast<f32x4*>(out + (long)m * ldo + n) = v;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();   // the next item's first store must not overtake a slower wave's last fragment reads
  }
}

template <int LAY_A, int LAY_B>
int launch_big_t(const GemmArgs& a, dim3 grid, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    TB_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_big_kernel<LAY_A, LAY_B>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const size_t lds_bytes = big_lds_bytes<LAY_A, LAY_B>();
  hipLaunchKernelGGL((gemm_big_kernel<LAY_A, LAY_B>), grid, dim3(kBigNT), lds_bytes, stream, a);
  return TB_OK;
}

int launch_big(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
  if (transA == 0 && transB == 0) return launch_big_t<0, 0>(a, grid, stream);
  if (transA == 0 && transB == 1) return launch_big_t<0, 1>(a, grid, stream);
  if (transA == 1 && transB == 1) return launch_big_t<1, 1>(a, grid, stream);
  return launch_big_t<1, 0>(a, grid, stream);
}

// Split-K reduction in a fixed order (bitwise reproducible), two levels so that it parallelises over the
// splits as well as over the elements: splitk_group sums groups of kGroup consecutive slabs into the
// first slab of each group; splitk_final sums the group leaders, adds the bias, applies the activation.
constexpr int kGroup = 16;

__global__ void splitk_group_kernel(float* __restrict__ slab, int splits, long total4) {
  const int g0 = blockIdx.y * kGroup;
  const int g1 = min(splits, g0 + kGroup);
  f32x4* s4 = reinterpret_cast<f32x4*>(slab);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    f32x4 a = s4[(size_t)g0 * total4 + i];
    for (int z = g0 + 1; z < g1; ++z) a += s4[(size_t)z * total4 + i];
    s4[(size_t)g0 * total4 + i] = a;
  }
}

// C[m][n] = act(sum over z = 0, zstep, 2*zstep, ... < splits of slab[z][m][n] + bias[n])
__global__ void splitk_final_kernel(const float* __restrict__ slab, int splits, int zstep, int M, int N, float* C,
                                    long ldc, const float* bias, const float* bias2, int act, int accumulate) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int z = 0; z < splits; z += zstep) s += slab[(size_t)z * total + i];
    const int m = (int)(i / N), n = (int)(i % N);
    if (bias || bias2) s += (bias ? bias[n] : 0.f) + (bias2 ? bias2[n] : 0.f);
    if (act == 1) s = tb_tanh(s);
    else if (act == 2) s = tb_sigmoid(s);
    if (accumulate) s += C[(long)m * ldc + n];
    C[(long)m * ldc + n] = s;
  }
}

// C[m][n] += tmp[m][n]: the accumulate form of a product that needed no k split
__global__ void accumulate_kernel(const float* __restrict__ tmp, int M, int N, float* C, long ldc) {
  const long total = (long)M * N;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int m = (int)(i / N), n = (int)(i % N);
    C[(long)m * ldc + n] += tmp[i];
  }
}

template <int BM, int BN, int WGM, int WGN, bool FAST>
void launch_lay(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
  // two k-tiles ahead where the registers allow it: the 128 x 128 tile (the 256-row tiles sit at 248-256 registers)
  constexpr int PF = (BM == 128 && BN == 128 && FAST) ? 2 : 1;
  if (transA == 0 && transB == 0) hipLaunchKernelGGL((gemm_bf16x3_kernel<0, 0, BM, BN, WGM, WGN, FAST, PF>), grid, dim3(256), 0, stream, a);
  else if (transA == 0 && transB == 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<0, 1, BM, BN, WGM, WGN, FAST, PF>), grid, dim3(256), 0, stream, a);
  else if (transA == 1 && transB == 1) hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 1, BM, BN, WGM, WGN, FAST, PF>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((gemm_bf16x3_kernel<1, 0, BM, BN, WGM, WGN, FAST, PF>), grid, dim3(256), 0, stream, a);
}

template <int BM, int BN, int WGM, int WGN>
void launch_cfg(const GemmArgs& a, int transA, int transB, dim3 grid, hipStream_t stream) {
  const bool out_vec = a.splits > 1 ? ((reinterpret_cast<uintptr_t>(a.slab) & 15) == 0 && (((size_t)a.M * a.N) & 3) == 0)
                                    : ((reinterpret_cast<uintptr_t>(a.C) & 15) == 0 && (a.ldc & 3) == 0);
  if (a.vecA && a.vecB && out_vec && (a.N & 3) == 0) launch_lay<BM, BN, WGM, WGN, true>(a, transA, transB, grid, stream);
  else launch_lay<BM, BN, WGM, WGN, false>(a, transA, transB, grid, stream);
}

}  // namespace

// Split-K slab bound: splits are chosen so that tiles x splits fills the chip twice over, and only when
// there are fewer tiles than CUs, so the slabs never exceed (3 x CUs) tiles of 128 x 128 floats.
size_t tb_gemm_ws_bytes() { return (size_t)3 * tb_num_cus() * 128 * 128 * sizeof(float); }

// Host-side launcher shared by the C-ABI entry points (capi.hip, lstm.hip).
// transA: 0 -> A is [M][K]; 1 -> A is [K][M].   transB: 0 -> B is [N][K]; 1 -> B is [K][N].
// accumulate: C += the product (weight gradients added straight into a gradient buffer); needs a workspace
// of at least M x N floats.
int tb_gemm_launch2(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                    long ldc, const float* bias, const float* bias2, int M, int N, int K, int act, int b_shift,
                    int b_period, float* ws, size_t ws_bytes, hipStream_t stream, int accumulate);

int tb_gemm_launch(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                   long ldc, const float* bias, int M, int N, int K, int act, int b_shift, int b_period,
                   float* ws, size_t ws_bytes, hipStream_t stream, int accumulate) {
  return tb_gemm_launch2(A, lda, transA, B, ldb, transB, C, ldc, bias, nullptr, M, N, K, act, b_shift, b_period, ws,
                         ws_bytes, stream, accumulate);
}

static int gemm_launch_one(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                           long ldc, const float* bias, const float* bias2, int M, int N, int K, int act, int b_shift,
                           int b_period, float* ws, size_t ws_bytes, hipStream_t stream, int accumulate,
                           TbGemmDeferred* defer = nullptr);

static bool gemm_fits(const float*, long lda, int transA, long ldb, int transB, int M, int N, int K) {
  constexpr long long kLimit = 0x7FFFFFF0ll / 4;   // floats
  const long long exA = transA ? (long long)(K - 1) * lda + M : (long long)(M - 1) * lda + K;
  const long long exB = transB ? (long long)(K - 1) * ldb + N : (long long)(N - 1) * ldb + K;
  return exA < kLimit && exB < kLimit;
}

int tb_gemm_launch_deferred(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C, long ldc,
                            int M, int N, int K, int b_shift, int b_period, float* ws, size_t ws_bytes, hipStream_t stream,
                            int accumulate, TbGemmDeferred* d) {
  d->splits = 0;
  if (M <= 0 || N <= 0) return TB_OK;
  if (K <= 0) return tb_set_error(TB_ERR_ARG, "gemm: K must be positive");
  if (!gemm_fits(A, lda, transA, ldb, transB, M, N, K))
    return tb_gemm_launch2(A, lda, transA, B, ldb, transB, C, ldc, nullptr, nullptr, M, N, K, 0, b_shift, b_period, ws,
                           ws_bytes, stream, accumulate);
  return gemm_launch_one(A, lda, transA, B, ldb, transB, C, ldc, nullptr, nullptr, M, N, K, 0, b_shift, b_period, ws,
                         ws_bytes, stream, accumulate, d);
}

// bias2: optional second bias vector; the epilogue adds (bias + bias2), the two summed first.
// The kernel addresses A and B through 32-bit buffer offsets (2 GiB each).  Larger operands -- the long-context
// shapes: 2048 rows x 1024 steps x 4H floats of gates are 17 GB -- are cut here: along M when A is [M][K] (the
// projections and dx: independent row blocks of C), along K when A is [K][M] (the weight gradients: K = rows x
// steps; later pieces accumulate into C), at multiples of the B-shift period so that no shifted row crosses a cut.
int tb_gemm_launch2(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                    long ldc, const float* bias, const float* bias2, int M, int N, int K, int act, int b_shift,
                    int b_period, float* ws, size_t ws_bytes, hipStream_t stream, int accumulate) {
  if (M <= 0 || N <= 0) return TB_OK;
  if (K <= 0) return tb_set_error(TB_ERR_ARG, "gemm: K must be positive");
  constexpr long long kLimit = 0x7FFFFFF0ll / 4;   // floats
  const long long exA = transA ? (long long)(K - 1) * lda + M : (long long)(M - 1) * lda + K;
  const long long exB = transB ? (long long)(K - 1) * ldb + N : (long long)(N - 1) * ldb + K;
  if (exA < kLimit && exB < kLimit)
    return gemm_launch_one(A, lda, transA, B, ldb, transB, C, ldc, bias, bias2, M, N, K, act, b_shift, b_period, ws,
                           ws_bytes, stream, accumulate);
  if (!transA) {
    if (exB >= kLimit) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: B operand larger than 2 GiB (%lld bytes)", exB * 4);
    if (b_shift) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: shifted B with a row-major A larger than 2 GiB");
    long long rows = (kLimit - K) / lda;
    if (rows >= 1024) rows = rows / 1024 * 1024;
    if (rows < 1) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: one row of A exceeds 2 GiB");
    for (long long m0 = 0; m0 < M; m0 += rows) {
      const int mc = (int)((M - m0) < rows ? (M - m0) : rows);
      const int rc = gemm_launch_one(A + m0 * lda, lda, 0, B, ldb, transB, C + m0 * ldc, ldc, bias, bias2, mc, N, K, act,
                                     0, b_period, ws, ws_bytes, stream, accumulate);
      if (rc) return rc;
    }
    return TB_OK;
  }
  // A is [K][M]: cut K; B must be cut the same way ([K][N])
  if (!transB) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: A^T larger than 2 GiB needs B as [K][N]");
  if (act || bias || bias2) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: K-cut product with an epilogue");
  const long period = b_period > 0 ? b_period : 1;
  long long kc = (kLimit - M) / lda;
  const long long kcB = (kLimit - N) / ldb;
  if (kcB < kc) kc = kcB;
  kc = kc / period * period;
  if (kc < 1) return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: one shift period of A^T exceeds 2 GiB");
  if (!ws || ws_bytes < (size_t)M * N * sizeof(float))
    return tb_set_error(TB_ERR_WORKSPACE, "gemm: a K-cut product needs a workspace of M x N floats");
  for (long long k0 = 0; k0 < K; k0 += kc) {
    const int kk = (int)((K - k0) < kc ? (K - k0) : kc);
    const int rc = gemm_launch_one(A + k0 * lda, lda, 1, B + k0 * ldb, ldb, 1, C, ldc, nullptr, nullptr, M, N, kk, 0,
                                   b_shift, b_period, ws, ws_bytes, stream, (k0 > 0 || accumulate) ? 1 : 0);
    if (rc) return rc;
  }
  return TB_OK;
}

static int gemm_launch_one(const float* A, long lda, int transA, const float* B, long ldb, int transB, float* C,
                           long ldc, const float* bias, const float* bias2, int M, int N, int K, int act, int b_shift,
                           int b_period, float* ws, size_t ws_bytes, hipStream_t stream, int accumulate,
                           TbGemmDeferred* defer) {
  if (M <= 0 || N <= 0) return TB_OK;
  if (accumulate && (!ws || ws_bytes < (size_t)M * N * sizeof(float)))
    return tb_set_error(TB_ERR_WORKSPACE, "gemm: accumulate needs a workspace of M x N floats");
  if (K <= 0) return tb_set_error(TB_ERR_ARG, "gemm: K must be positive");
  GemmArgs a;
  a.A = A; a.B = B; a.C = C; a.bias = bias; a.bias2 = bias2;
  a.M = M; a.N = N; a.K = K;
  a.lda = lda; a.ldb = ldb; a.ldc = ldc;
  a.act = act;
  a.b_shift = b_shift;
  a.b_period = b_period > 0 ? b_period : 1;
  a.vecA = ((reinterpret_cast<uintptr_t>(A) & 15) == 0 && (lda & 3) == 0) ? 1 : 0;
  a.vecB = ((reinterpret_cast<uintptr_t>(B) & 15) == 0 && (ldb & 3) == 0) ? 1 : 0;
  // the kernel addresses the operands through 32-bit buffer offsets
  const long long exA = (transA ? (long long)(K - 1) * lda + M : (long long)(M - 1) * lda + K) * 4;
  const long long exB = (transB ? (long long)(K - 1) * ldb + N : (long long)(N - 1) * ldb + K) * 4;
  if (exA >= 0x7FFFFFF0ll || exB >= 0x7FFFFFF0ll)
    return tb_set_error(TB_ERR_UNSUPPORTED, "gemm: operand larger than 2 GiB (%lld / %lld bytes)", exA, exB);
  a.bytesA = (int)exA;
  a.bytesB = (int)exB;
  static const int xcd_map = []() { const char* e = getenv("TB_GEMM_XMAP"); return e ? atoi(e) : 1; }();
  a.xcd_map = xcd_map;
  // tile shape by output width
  // TB_GEMM_BIG (default 1): the 256 x 128 / 512-thread kernel for long k loops over wide outputs (gemm_big_kernel)
  static const int big_on = []() { const char* e = getenv("TB_GEMM_BIG"); return e ? atoi(e) : 1; }();
  const bool out_vec_big = ((reinterpret_cast<uintptr_t>(ws) & 15) == 0) && ((reinterpret_cast<uintptr_t>(C) & 15) == 0) &&
                           (ldc & 3) == 0 && (N & 3) == 0 && (((size_t)M * N) & 3) == 0;
  const bool big = big_on && N > 64 && M >= 256 && K >= 512 && a.vecA && a.vecB && out_vec_big;
  const int cfg = big ? 3 : (N <= 32 ? 0 : (N <= 64 ? 1 : 2));
  const int BM = cfg == 2 ? 128 : 256, BN = cfg == 0 ? 32 : (cfg == 1 ? 64 : 128);
  const int mt = (M + BM - 1) / BM, nt = (N + BN - 1) / BN;
  const long tiles = (long)mt * nt;
  int splits = 1;
  const int cus = tb_num_cus();
  if (tiles < cus && K >= 32 * BK && ws) {
    splits = (int)(((big ? 1L : 2L) * cus + tiles - 1) / tiles);
    const int maxs = K / (8 * BK);
    if (splits > maxs) splits = maxs;
    if (splits > 512) splits = 512;
    while (splits > 1 && (size_t)splits * M * N * sizeof(float) > ws_bytes) --splits;
    if (splits < 1) splits = 1;
  }
  int kchunk = (K + splits - 1) / splits;
  kchunk = (kchunk + BK - 1) / BK * BK;
  splits = (K + kchunk - 1) / kchunk;
  a.kchunk = kchunk;
  a.splits = splits;
  a.slab = ws;
  float* const Cout = C;
  const long ldout = ldc;
  if (accumulate && splits == 1) {   // product into the workspace, added to C below
    a.C = ws;
    a.ldc = N;
  }
  // persistent grid: as many blocks as stay resident (2 per CU by registers, 3 for the narrow tiles)
  long items = tiles * splits;
  const long resident = (long)cus * (cfg == 0 ? 3 : (cfg == 3 ? 1 : 2));
  dim3 grid((unsigned)(items < resident ? items : resident));
  tb_prof_begin(TB_K_GEMM, stream);
  if (cfg == 3) {
    const int rc = launch_big(a, transA, transB, grid, stream);
    if (rc) return rc;
  } else if (cfg == 0) launch_cfg<256, 32, 4, 1>(a, transA, transB, grid, stream);
  else if (cfg == 1) launch_cfg<256, 64, 4, 1>(a, transA, transB, grid, stream);
  else launch_cfg<128, 128, 2, 2>(a, transA, transB, grid, stream);
  TB_LAUNCH_CHECK("gemm_bf16x3_kernel");
  if (defer && (splits > 1 || accumulate)) {   // the caller sums the slabs (and adds to C) in a kernel of its own
    defer->slab = ws; defer->splits = splits; defer->M = M; defer->N = N; defer->accumulate = accumulate;
    defer->C = Cout; defer->ldc = ldout;
    tb_prof_end(TB_K_GEMM, stream);
    return TB_OK;
  }
  if (splits > 1) {
    const long total = (long)M * N;
    int zstep = 1;
    if (splits > kGroup && (total & 3) == 0 && (reinterpret_cast<uintptr_t>(ws) & 15) == 0) {
      int bx = (int)((total / 4 + 255) / 256);
      if (bx > 1024) bx = 1024;
      hipLaunchKernelGGL(splitk_group_kernel, dim3(bx, (splits + kGroup - 1) / kGroup), dim3(256), 0, stream, ws, splits,
                         total / 4);
      TB_LAUNCH_CHECK("splitk_group_kernel");
      zstep = kGroup;
    }
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_final_kernel, dim3(blocks), dim3(256), 0, stream, ws, splits, zstep, M, N, C, ldc, bias,
                       bias2, act, accumulate);
    TB_LAUNCH_CHECK("splitk_final_kernel");
  } else if (accumulate) {
    int blocks = (int)(((long)M * N + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(accumulate_kernel, dim3(blocks), dim3(256), 0, stream, ws, M, N, Cout, ldout);
    TB_LAUNCH_CHECK("accumulate_kernel");
  }
  tb_prof_end(TB_K_GEMM, stream);
  return TB_OK;
}
