// fp32 "NT" GEMM on the CDNA4 matrix cores:  C = act(((A . W^T) / divisor) * colscale + bias) + residual
//   A (M,K) row-major (lda), W (N,K) row-major (ldw, torch nn.Linear weight layout), C (M,N) (ldc), optional batch.
// Serves every dense projection of the matching path (nn.Linear / 1x1 conv call sites: PEM/model/transformer.py:127-129,
// 186-188, 390-393, 548-550; PEM/model/coarse_point_matching.py:35-38; PEM/model/fine_point_matching.py:47-51) and the
// feature-similarity contraction (PEM/utils/model_utils.py:144).
//
// v_mfma_f32_32x32x2_f32: exact f32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD.  Block tile 128x128x16 (or 64x64x16),
// 4 waves in a 2x2 grid.  Operands are staged in LDS with an odd row stride (17 dwords) so the ds_read_b32 fragment
// reads (32 different rows per lane group) are conflict-free; the MFMA issue time (64 cycles each) dominates, several
// blocks per CU plus the register prefetch of the next K tile hide the staging.
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GM_BK 16
#define GM_LD 17

// BM x BN block tile, 4 waves in a 2x2 grid, each wave (BM/2)x(BN/2) = TM x TN MFMA tiles.  128x128 for the big
// problems; 64x64 when the 128-tile grid would not fill the 256 CUs (the 197-token sparse layers: M = 6 304 / 12 608).
// The next K tile is fetched into registers while the current one is multiplied (global latency hidden behind 16-32
// MFMAs per wave).
// second (inner) batch level: workgroup z = b1 * n2 + b2 addresses  base + b1 * s + b2 * s2   (n2 == 1: plain batch)
struct Batch2 {
  int n2;
  long sA2, sW2, sC2, sR2;
  int fold_nz;  // > 0: the batch elements are folded into grid.x (gemm_tile below), value = number of elements
};

// Workgroup -> (batch element z, row tile, column tile).  Workgroup ids are dealt round-robin to the 8 XCDs, each with its
// own L2.
//  * One big problem (tiles_n <= 8, tiles_m >= 32): the tiles_n column tiles of one row tile get ids 8 apart (same XCD,
//    dispatched together), so the A rows they share come from HBM once and from that XCD's L2 afterwards; eight
//    consecutive row tiles form a group of 8 * tiles_n ids (the last group is padded).
//  * Batches of 8 or more (fold_nz): element z's tiles get the ids congruent to z mod 8, i.e. ONE XCD reads that element's
//    A and W (with the elements on grid.z every XCD's L2 fetched every element: the 32 x 2049 x 2049 similarity GEMM pulled
//    1.4 GB from HBM for 134 MB of operands, rocprof FETCH_SIZE).  Groups of 8 elements, the last group padded.
template <int BM, int BN>
__device__ __forceinline__ bool gemm_tile(int M, int N, const Batch2& b2, int& z, int& tm_, int& tn_) {
  const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
  if (b2.fold_nz > 0) {
    const int per = tiles_m * tiles_n;
    const int g = blockIdx.x / (8 * per), r = blockIdx.x % (8 * per);
    z = g * 8 + (r & 7);
    if (z >= b2.fold_nz) return false;
    const int x = r >> 3;
    tm_ = x % tiles_m;
    tn_ = x / tiles_m;
    return true;
  }
  z = blockIdx.z;
  if (tiles_n <= 8 && tiles_m >= 32) {  // (small problems: plain order, no padding, every XCD gets tiles)
    const int g = blockIdx.x / (8 * tiles_n), r = blockIdx.x % (8 * tiles_n);
    tm_ = g * 8 + (r & 7);
    tn_ = r >> 3;
    return tm_ < tiles_m;  // padding of the last group (uniform for the workgroup)
  }
  tm_ = blockIdx.x % tiles_m;
  tn_ = blockIdx.x / tiles_m;
  return true;
}

// Exact fp32 main loop of one BM x BN tile (v_mfma_f32_32x32x2_f32: a k-ordered fmaf chain).  Shared by gemm_nt_kernel and by the
// split-precision kernel's out-of-range fallback.  As / Bs: BM x GM_LD and BN x GM_LD floats of LDS.
template <int BM, int BN>
__device__ __forceinline__ void gemm_exact_mainloop(f32x16 (&acc)[BM / 64][BN / 64], const float* __restrict__ A,
                                                    const float* __restrict__ W, int M, int N, int K, long lda, long ldw, int m0,
                                                    int n0, float* __restrict__ As, float* __restrict__ Bs) {
  constexpr int TM = BM / 64, TN = BN / 64;  // MFMA tiles per wave
  constexpr int RA = BM / 64, RB = BN / 64;  // float4 staging loads per thread
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging map: rows (t>>2) + 64*u, k4 = (t&3)*4
  const int sr = t >> 2, sk = (t & 3) * 4;
  const float* ap[RA];
  const float* bp[RB];
#pragma unroll
  for (int u = 0; u < RA; ++u) ap[u] = A + (size_t)min(m0 + sr + 64 * u, M - 1) * lda + sk;
#pragma unroll
  for (int u = 0; u < RB; ++u) bp[u] = W + (size_t)min(n0 + sr + 64 * u, N - 1) * ldw + sk;
  const bool vec = ((lda & 3) == 0) && ((ldw & 3) == 0) && ((((size_t)A | (size_t)W) & 15) == 0);

  float4 va[RA], vb[RB];
  auto fetch = [&](int k0) {
    if (vec && k0 + GM_BK <= K) {
#pragma unroll
      for (int u = 0; u < RA; ++u) va[u] = *reinterpret_cast<const float4*>(ap[u] + k0);
#pragma unroll
      for (int u = 0; u < RB; ++u) vb[u] = *reinterpret_cast<const float4*>(bp[u] + k0);
    } else {
      float tmp[4];
#pragma unroll
      for (int u = 0; u < RA; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = (k0 + sk + e) < K ? ap[u][k0 + e] : 0.f;
        va[u] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
      }
#pragma unroll
      for (int u = 0; u < RB; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = (k0 + sk + e) < K ? bp[u][k0 + e] : 0.f;
        vb[u] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
      }
    }
  };

  const int fr = lane & 31, fk = lane >> 5;
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += GM_BK) {
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      float* d = As + (sr + 64 * u) * GM_LD + sk;
      d[0] = va[u].x; d[1] = va[u].y; d[2] = va[u].z; d[3] = va[u].w;
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      float* d = Bs + (sr + 64 * u) * GM_LD + sk;
      d[0] = vb[u].x; d[1] = vb[u].y; d[2] = vb[u].z; d[3] = vb[u].w;
    }
    __syncthreads();
    if (k0 + GM_BK < K) fetch(k0 + GM_BK);  // in flight during the MFMAs below
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[(wm + 32 * i + fr) * GM_LD + kk + fk];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[(wn + 32 * j + fr) * GM_LD + kk + fk];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
  }
}

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                      const float* __restrict__ bias, const float* __restrict__ colscale,
                                                      const float* __restrict__ residual, float* __restrict__ C, int M, int N,
                                                      int K, long lda, long ldw, long ldc, long ldr, long sA, long sW, long sC,
                                                      long sR, float divisor, int act, Batch2 b2) {
  constexpr int TM = BM / 64, TN = BN / 64;  // MFMA tiles per wave
  __shared__ float As[BM * GM_LD];
  __shared__ float Bs[BN * GM_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int zz, tm_, tn_;
  if (!gemm_tile<BM, BN>(M, N, b2, zz, tm_, tn_)) return;  // padding workgroup (uniform)
  const int bz = zz / b2.n2, bi = zz % b2.n2;
  A += (size_t)bz * sA + (size_t)bi * b2.sA2;
  W += (size_t)bz * sW + (size_t)bi * b2.sW2;
  C += (size_t)bz * sC + (size_t)bi * b2.sC2;
  if (residual) residual += (size_t)bz * sR + (size_t)bi * b2.sR2;
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

  f32x16 acc[TM][TN];
  gemm_exact_mainloop<BM, BN>(acc, A, W, M, N, K, lda, ldw, m0, n0, As, Bs);
  const int fr = lane & 31, fk = lane >> 5;
  // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + j * 32 + fr;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      const float cs = colscale ? colscale[col] : 1.f;
      // all 16 residual loads first: C may alias `residual` (in-place add), so the compiler must not be left to
      // interleave each load behind the previous store (that serialised 64 memory round trips per lane)
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        rv[r] = (residual && row < M) ? residual[(size_t)row * ldr + col] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < M) {
          float v = acc[i][j][r];
          if (divisor != 1.0f) v = v / divisor;
          v = colscale ? fmaf(v, cs, bv) : v + bv;
          if (act == 1) v = v > 0.f ? v : 0.f;
          v += rv[r];
          C[(size_t)row * ldc + col] = v;
        }
      }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Split-precision variant ("fp16 x3"): every fp32 operand x is cut into x_hi = fp16(x), x_lo = fp16(x - x_hi)
// (22 significand bits together) and a.b is evaluated as a_lo.b_hi + a_hi.b_lo + a_hi.b_hi on
// v_mfma_f32_32x32x16_f16: fp16 products are exact in the fp32 accumulator, the dropped a_lo.b_lo term is 2^-22
// relative.  Three MFMAs at 16x the fp32-MFMA rate = 5.3x the exact kernel's ceiling at ~1e-6 relative error
// (tests compare both modes with the same golden vectors).  The split happens while staging the tile into LDS.
// LDS rows are 40 halves (80 B): 16 consecutive rows land on 16 different 16-byte slots of the 256-B bank row, so
// the ds_read_b128 fragment reads are conflict-free.
// ---------------------------------------------------------------------------------------------------------------
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
#define H_BK 32
#define H_LD 40

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split4(const float4 v, half4& hi, half4& lo) {
  unsigned h0, h1, l0, l1;
  sam6d_split2_f16(v.x, v.y, h0, l0);
  sam6d_split2_f16(v.z, v.w, h1, l1);
  hi = __builtin_bit_cast(half4, u32x2{h0, h1});
  lo = __builtin_bit_cast(half4, u32x2{l0, l1});
}

// WS: the weight operand arrives pre-split -- Wh / Wl = fp16 hi / lo of W * 2^e, same (N, K) layout and strides as W, cut once at
// weight-load time (sam6d_split_f16) -- so staging it is a copy: the per-k-step VALU split of the weight tile (half of the
// kernel's vector work at K = 256) disappears, and so does its range check (the pack scale puts max |W| into [2^13, 2^14)).
// FAST: whole tiles (M % BM == 0 per batch entry, N % BN == 0, K % 32 == 0), 16-byte aligned operands, three products, the wide
// epilogue without divisor / column scale / activation -- every guard of the general form is then a compile-time constant.
template <int BM, int BN, bool WS, bool FAST = false>
__global__ __launch_bounds__(256, FAST ? 3 : 2) void gemm_nt_h3_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ colscale,
                                                         const float* __restrict__ residual, float* __restrict__ C, int M,
                                                         int N, int K, long lda, long ldw, long ldc, long ldr, long sA, long sW,
                                                         long sC, long sR, float divisor_, int act_, Batch2 b2, int wide_, int half_,
                                                         const _Float16* __restrict__ Wh, const _Float16* __restrict__ Wl,
                                                         float w_unscale) {
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int RA = BM / 32, RB = BN / 32;  // float4 staging loads per thread (32 rows x 8 float4 per pass)
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (BM + BN) * H_LD];  // also the epilogue's transpose slabs
  _Float16* Ah = smem;
  _Float16* Al = Ah + BM * H_LD;
  _Float16* Bh = Al + BM * H_LD;
  _Float16* Bl = Bh + BN * H_LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wide = FAST ? 1 : wide_, half = FAST ? 0 : half_, act = FAST ? 0 : act_;
  const float divisor = FAST ? 1.0f : divisor_;
  if (FAST) colscale = nullptr;
  int zz, tm_, tn_;
  if (!gemm_tile<BM, BN>(M, N, b2, zz, tm_, tn_)) return;  // padding workgroup (uniform)
  const int bz = zz / b2.n2, bi = zz % b2.n2;
  A += (size_t)bz * sA + (size_t)bi * b2.sA2;
  W += (size_t)bz * sW + (size_t)bi * b2.sW2;
  if (WS) {
    Wh += (size_t)bz * sW + (size_t)bi * b2.sW2;
    Wl += (size_t)bz * sW + (size_t)bi * b2.sW2;
  }
  C += (size_t)bz * sC + (size_t)bi * b2.sC2;
  if (residual) residual += (size_t)bz * sR + (size_t)bi * b2.sR2;
  const int m0 = tm_ * BM, n0 = tn_ * BN;
  const int wm = (wave >> 1) * (BM / 2), wn = (wave & 1) * (BN / 2);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int sr = t >> 3, sk = (t & 7) * 4;
  const float* ap[RA];
  const float* bp[RB];
#pragma unroll
  for (int u = 0; u < RA; ++u) ap[u] = A + (size_t)(FAST ? m0 + sr + 32 * u : min(m0 + sr + 32 * u, M - 1)) * lda + sk;
#pragma unroll
  for (int u = 0; u < RB; ++u) bp[u] = W + (size_t)(FAST ? n0 + sr + 32 * u : min(n0 + sr + 32 * u, N - 1)) * ldw + sk;
  const bool vec = FAST || (((lda & 3) == 0) && ((ldw & 3) == 0) && ((((size_t)A | (size_t)W) & 15) == 0));

  // Register ring of NS k-steps for the A rows (round 4): the loads of step i + NS - 1 are issued while step i is multiplied.  A is the
  // HBM stream (each row is read once); the weight tile comes from L2 (every workgroup re-reads it) and keeps its single step in flight.
  // Measured on the 65536 x 256 x 256 projections of the fine stage (scratch/ub_gemm.py, same bits): in_proj 57.3 -> 54.8 us, mlp3
  // (+ residual) 66.5 -> 61.6 us.  (A ring over BOTH operands needed 256 VGPRs + 48 B of scratch and was slower: 65.9 / 75.7 us.)
  // The kernel is not latency-bound at K = 256 as first assumed: rocprof counts ~3700 vector instructions per wave and tile (40 % of the
  // SIMDs' issue cycles) against 128 MFMAs (16 %) -- staging, splitting and the epilogue per byte moved.
  constexpr int NS = FAST ? 2 : 3;  // (FAST runs three workgroups per CU: 168 registers)
  float4 va[NS][RA], vb[RB];
  half4 wbh[RB], wbl[RB];
  const bool wvec = FAST || (WS && ((ldw & 3) == 0) && ((((size_t)Wh | (size_t)Wl) & 7) == 0));
  auto fetch_w16 = [&](int k0) {  // pre-split weight rows: 8-byte loads of 4 halves
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      const size_t o = (size_t)(FAST ? n0 + sr + 32 * u : min(n0 + sr + 32 * u, N - 1)) * ldw + sk + k0;
      if (wvec && (FAST || k0 + H_BK <= K)) {
        wbh[u] = *reinterpret_cast<const half4*>(Wh + o);
        wbl[u] = *reinterpret_cast<const half4*>(Wl + o);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool ok = (k0 + sk + e) < K;
          wbh[u][e] = ok ? Wh[o + e] : (_Float16)0.f;
          wbl[u][e] = ok ? Wl[o + e] : (_Float16)0.f;
        }
      }
    }
  };
  auto fetch_a = [&](float4 (&fa)[RA], int k0) {
    if (vec && (FAST || k0 + H_BK <= K)) {
#pragma unroll
      for (int u = 0; u < RA; ++u) fa[u] = *reinterpret_cast<const float4*>(ap[u] + k0);
    } else {
      float tmp[4];
#pragma unroll
      for (int u = 0; u < RA; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = (k0 + sk + e) < K ? ap[u][k0 + e] : 0.f;
        fa[u] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
      }
    }
  };
  auto fetch_w = [&](int k0) {
    if (WS) {
      fetch_w16(k0);
    } else if (vec && (FAST || k0 + H_BK <= K)) {
#pragma unroll
      for (int u = 0; u < RB; ++u) vb[u] = *reinterpret_cast<const float4*>(bp[u] + k0);
    } else {
      float tmp[4];
#pragma unroll
      for (int u = 0; u < RB; ++u) {
#pragma unroll
        for (int e = 0; e < 4; ++e) tmp[e] = (k0 + sk + e) < K ? bp[u][k0 + e] : 0.f;
        vb[u] = make_float4(tmp[0], tmp[1], tmp[2], tmp[3]);
      }
    }
  };

  const int fr = lane & 31, fk = lane >> 5;
  float ma = 0.f, mw = 0.f;  // running max |A|, max |W| of what this thread stages (range check below)
  fetch_w(0);
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st * H_BK < K) fetch_a(va[st], st * H_BK);
  for (int kb = 0; kb < K; kb += NS * H_BK) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int k0 = kb + st * H_BK;
      if (k0 < K) {  // (uniform)
        __syncthreads();
#pragma unroll
        for (int u = 0; u < RA; ++u) {
          const float4 v = va[st][u];
          ma = fmaxf(fmaxf(ma, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
          half4 hi, lo;
          split4(v, hi, lo);
          *reinterpret_cast<half4*>(&Ah[(sr + 32 * u) * H_LD + sk]) = hi;
          if (!half) *reinterpret_cast<half4*>(&Al[(sr + 32 * u) * H_LD + sk]) = lo;
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          half4 hi, lo;
          if (WS) {
            hi = wbh[u];
            lo = wbl[u];
          } else {
            mw = fmaxf(fmaxf(mw, fabsf(vb[u].x)), fmaxf(fabsf(vb[u].y), fmaxf(fabsf(vb[u].z), fabsf(vb[u].w))));
            split4(vb[u], hi, lo);
          }
          *reinterpret_cast<half4*>(&Bh[(sr + 32 * u) * H_LD + sk]) = hi;
          if (!half) *reinterpret_cast<half4*>(&Bl[(sr + 32 * u) * H_LD + sk]) = lo;
        }
        __syncthreads();
        if (k0 + H_BK < K) fetch_w(k0 + H_BK);
        if (k0 + (NS - 1) * H_BK < K) fetch_a(va[(st + NS - 1) % NS], k0 + (NS - 1) * H_BK);
        if (half) {  // single product: hi halves only (workgroup-uniform branch)
#pragma unroll
          for (int ks = 0; ks < H_BK; ks += 16) {
            half8 ah[TM], bh[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ah[i] = *reinterpret_cast<const half8*>(&Ah[(wm + 32 * i + fr) * H_LD + ks + 8 * fk]);
#pragma unroll
            for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const half8*>(&Bh[(wn + 32 * j + fr) * H_LD + ks + 8 * fk]);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
        } else {
#pragma unroll
          for (int ks = 0; ks < H_BK; ks += 16) {
            half8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
              ah[i] = *reinterpret_cast<const half8*>(&Ah[(wm + 32 * i + fr) * H_LD + ks + 8 * fk]);
              al[i] = *reinterpret_cast<const half8*>(&Al[(wm + 32 * i + fr) * H_LD + ks + 8 * fk]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              bh[j] = *reinterpret_cast<const half8*>(&Bh[(wn + 32 * j + fr) * H_LD + ks + 8 * fk]);
              bl[j] = *reinterpret_cast<const half8*>(&Bl[(wn + 32 * j + fr) * H_LD + ks + 8 * fk]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
              }
          }
        }
      }
    }
  }
  // ---- range of the fp16 split.  x = hi + lo carries 22 significand bits only while hi cannot overflow (|x| < 65520) and lo is not
  // pushed far into fp16's subnormals (|x| well above 2^-14 * 2^11).  A tile whose operands leave [2^-6, 2^15) -- raw features of an
  // unknown checkpoint, unnormalised descriptors -- is recomputed here with the exact fp32 MFMA loop of gemm_nt_kernel, which has
  // fp32's own range; in-range tiles (everything the matching path produces itself) are untouched, so results stay independent of the
  // batch neighbours that share a tile.  All-zero operands are left alone.
  {
    __shared__ float red[8];  // (its own words: no barrier needed against the staging buffers the other waves may still read)
    ma = wave_max_dpp(ma);
    mw = wave_max_dpp(mw);
    if (lane == 0) { red[wave] = ma; red[4 + wave] = mw; }
    __syncthreads();
    const float ta = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float tw = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
    const bool bad = !(ta < 32768.0f) || !(tw < 32768.0f) || (ta > 0.f && ta < 0.015625f) || (tw > 0.f && tw < 0.015625f);
    if (bad) {  // (uniform for the workgroup; the exact loop and the epilogue put their own barriers before they touch the LDS)
      float* As = reinterpret_cast<float*>(smem);
      gemm_exact_mainloop<BM, BN>(acc, A, W, M, N, K, lda, ldw, m0, n0, As, As + BM * GM_LD);
      __syncthreads();
    } else if (WS) {  // undo the weight pack scale (a power of two: exact)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] *= w_unscale;
    }
  }
  if (wide) {
    // Wide epilogue: each wave transposes its 32 x (BN/2) accumulator slab through LDS so that a lane owns 4
    // consecutive columns of one row: residual loads and C stores become 16-byte accesses of full 128/256-byte row
    // segments (the accumulator layout itself gives 4-byte accesses, 4x the instructions and half-used lines).
    // (at most two 32-column tiles per pass: a 64 x 256 workgroup tile goes through the slab in two passes per row tile)
    constexpr int EJ = TN > 2 ? 2 : TN, WC = EJ * 32, SLD = WC + 4, LPR = WC / 4, RPP = 64 / LPR, NP = 32 / RPP;
    __syncthreads();  // every wave is done reading operand fragments: the staging buffers can be reused
    float* slab = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
    const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jp = 0; jp < TN / EJ; ++jp) {
      const int col = n0 + wn + jp * WC + c4;
      float4 bv4 = make_float4(0.f, 0.f, 0.f, 0.f), cs4 = make_float4(1.f, 1.f, 1.f, 1.f);
      if (FAST || col < N) {
        if (bias) bv4 = *reinterpret_cast<const float4*>(bias + col);
        if (colscale) cs4 = *reinterpret_cast<const float4*>(colscale + col);
      }
#pragma unroll
      for (int j = 0; j < EJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * fk) * SLD + j * 32 + fr] = acc[i][jp * EJ + j][r];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      float4 rv[NP];
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int row = m0 + wm + i * 32 + it * RPP + rr0;
        rv[it] = (residual && (FAST || (row < M && col < N))) ? *reinterpret_cast<const float4*>(residual + (size_t)row * ldr + col)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int it = 0; it < NP; ++it) {
        const int rr = it * RPP + rr0;
        const int row = m0 + wm + i * 32 + rr;
        if (FAST || (row < M && col < N)) {
          float4 v = *reinterpret_cast<const float4*>(&slab[rr * SLD + c4]);
          float e[4] = {v.x, v.y, v.z, v.w};
          const float bb[4] = {bv4.x, bv4.y, bv4.z, bv4.w}, cc[4] = {cs4.x, cs4.y, cs4.z, cs4.w};
          const float rz[4] = {rv[it].x, rv[it].y, rv[it].z, rv[it].w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float x = e[q];
            if (divisor != 1.0f) x = x / divisor;
            x = colscale ? fmaf(x, cc[q], bb[q]) : x + bb[q];
            if (act == 1) x = x > 0.f ? x : 0.f;
            e[q] = x + rz[q];
          }
          *reinterpret_cast<float4*>(C + (size_t)row * ldc + col) = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      __builtin_amdgcn_wave_barrier();  // slab is rewritten by the next row tile
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn + j * 32 + fr;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      const float cs = colscale ? colscale[col] : 1.f;
      // all 16 residual loads first: C may alias `residual` (in-place add), so the compiler must not be left to
      // interleave each load behind the previous store (that serialised 64 memory round trips per lane)
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        rv[r] = (residual && row < M) ? residual[(size_t)row * ldr + col] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < M) {
          float v = acc[i][j][r];
          if (divisor != 1.0f) v = v / divisor;
          v = colscale ? fmaf(v, cs, bv) : v + bv;
          if (act == 1) v = v > 0.f ? v : 0.f;
          v += rv[r];
          C[(size_t)row * ldc + col] = v;
        }
      }
    }
}

// 0 = exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = fp16 x3 split (default), 2 = fp16 single product (a_hi . b_hi only, fp32
// accumulate: the reduced-precision variant of BASELINE.json config 5, ~1e-3 relative).  Process-wide; set before launching.
static int g_matmul_mode = 1;
// The calling thread's override of the process default (-1 = none): a host that runs two weight sets with different arithmetic in one
// process brackets each call sequence with sam6d_set_thread_matmul_mode(mode) ... (-1) instead of flipping the process default.
static thread_local int t_matmul_mode = -1;
extern "C" int sam6d_set_matmul_mode(int mode) {
  SAM6D_REQUIRE(mode == 0 || mode == 1 || mode == 2, "set_matmul_mode: 0 (exact fp32 MFMA), 1 (fp16 x3 split) or 2 (fp16 single product)");
  g_matmul_mode = mode;
  return 0;
}
extern "C" int sam6d_set_thread_matmul_mode(int mode) {
  SAM6D_REQUIRE(mode >= -1 && mode <= 2, "set_thread_matmul_mode: -1 (follow the process default), 0, 1 or 2");
  t_matmul_mode = mode;
  return 0;
}
extern "C" int sam6d_get_matmul_mode(void) { return t_matmul_mode >= 0 ? t_matmul_mode : g_matmul_mode; }
extern "C" int sam6d_get_thread_matmul_mode(void) { return t_matmul_mode; }

static bool gemm_fast_enabled() {  // SAM6D_GEMM_FAST=0: the general kernel also for whole-tile launches (A/B runs)
  static int on = -1;
  if (on < 0) {
    const char* e = getenv("SAM6D_GEMM_FAST");
    on = (e && e[0] == '0') ? 0 : 1;
  }
  return on != 0;
}

static int gemm_launch(const float* A, const float* W, const float* bias, const float* colscale, const float* residual,
                       float* C, int M, int N, int K, long lda, long ldw, long ldc, long ldr, int batch, long sA, long sW,
                       long sC, long sR, Batch2 b2, float divisor, int act, void* stream, const void* Wh = nullptr,
                       const void* Wl = nullptr, float w_scale = 1.0f) {
  SAM6D_REQUIRE(A && W && C, "gemm_nt: null pointer");
  SAM6D_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0 && b2.n2 >= 1, "gemm_nt: bad sizes M=%d N=%d K=%d batch=%d x %d", M, N,
                K, batch, b2.n2);
  SAM6D_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm_nt: leading dimension smaller than the row length");
  const int keep_split = act & 16;  // geometric operand: stays at fp16 x3 in matmul mode 2 (sam6d_hip.h)
  act &= ~16;
  SAM6D_REQUIRE(act == 0 || act == 1, "gemm_nt: act must be 0 (none) or 1 (ReLU), optionally + 16");
  SAM6D_REQUIRE((long)batch * b2.n2 <= 65535, "gemm_nt: batch (x batch2) must be <= 65535");
  if (M == 0 || N == 0 || batch == 0) return 0;
  const int nz = batch * b2.n2;
  const long blocks128 = (long)cdiv(M, 128) * cdiv(N, 128) * nz;
  // big tiles when they fill the chip (>= 4 workgroups per CU) and do not mostly pad (M = 197 would waste 42 % of a 2 x 128 split)
  const bool big = blocks128 >= 1024 && (M > 256 || M % 128 == 0);
  // (64 x 256 tiles -- the whole output row in one workgroup, every A row read and split once -- were measured on the fine in_proj,
  // M = 131 136, N = K = 256 with pre-split weights: 128 us against 115 us for 128 x 128 tiles; the epilogue keeps its two-pass form)
  const int bm = big ? 128 : 64, bn = big ? 128 : 64;
  const int tm = cdiv(M, bm), tn = cdiv(N, bn);
  // all tiles on x (2^31 limit) in the XCD-aware order the kernels decode; groups of 8 row tiles are padded
  const bool fold = nz >= 8;  // batch elements folded into grid.x, one XCD per element (gemm_tile)
  const long tiles = fold ? (long)cdiv(nz, 8) * 8 * tm * tn : (tn <= 8 && tm >= 32) ? (long)cdiv(tm, 8) * 8 * tn : (long)tm * tn;
  SAM6D_REQUIRE(tiles < 2147483647L, "gemm_nt: too many tiles for one launch");
  dim3 grid((unsigned)tiles, 1, fold ? 1 : nz);
  b2.fold_nz = fold ? nz : 0;
  hipStream_t st = (hipStream_t)stream;
#define GEMM_LAUNCH(KERNEL, ...)                                                                                         \
  hipLaunchKernelGGL(KERNEL, grid, dim3(256), 0, st, A, W, bias, colscale, residual, C, M, N, K, lda, ldw, ldc, ldr, sA, sW, \
                     sC, sR, divisor, act, b2, ##__VA_ARGS__)
  if (sam6d_get_matmul_mode() >= 1 && K >= 32) {
    const int half = (sam6d_half_for(0) && !keep_split) ? 1 : 0;
    // 16-byte epilogue accesses need 4-float alignment of every row start and of the per-column vectors
    const int wide = ((N & 3) == 0 && (ldc & 3) == 0 && (sC & 3) == 0 && (b2.sC2 & 3) == 0 && (((size_t)C) & 15) == 0 &&
                      (!residual || ((ldr & 3) == 0 && (sR & 3) == 0 && (b2.sR2 & 3) == 0 && (((size_t)residual) & 15) == 0)) &&
                      (!bias || (((size_t)bias) & 15) == 0) && (!colscale || (((size_t)colscale) & 15) == 0))
                         ? 1 : 0;
    const _Float16* wh = reinterpret_cast<const _Float16*>(Wh);
    const _Float16* wl = reinterpret_cast<const _Float16*>(Wl);
    const float wu = 1.0f / w_scale;
    if (wh && wl) {
      const bool fast = big && wide && !half && (M % 128) == 0 && (N % 128) == 0 && (K % 32) == 0 && (lda & 3) == 0 && (ldw & 3) == 0 &&
                        ((((size_t)A | (size_t)W) & 15) == 0) && ((((size_t)Wh | (size_t)Wl) & 7) == 0) && (sA & 3) == 0 && (b2.sA2 & 3) == 0 &&
                        (sW & 3) == 0 && (b2.sW2 & 3) == 0 && divisor == 1.0f && !colscale && act == 0 && gemm_fast_enabled();
      if (fast) GEMM_LAUNCH((gemm_nt_h3_kernel<128, 128, true, true>), wide, half, wh, wl, wu);
      else if (big) GEMM_LAUNCH((gemm_nt_h3_kernel<128, 128, true>), wide, half, wh, wl, wu);
      else GEMM_LAUNCH((gemm_nt_h3_kernel<64, 64, true>), wide, half, wh, wl, wu);
    } else {
      if (big) GEMM_LAUNCH((gemm_nt_h3_kernel<128, 128, false>), wide, half, wh, wl, wu);
      else GEMM_LAUNCH((gemm_nt_h3_kernel<64, 64, false>), wide, half, wh, wl, wu);
    }
  } else {
    if (big) GEMM_LAUNCH((gemm_nt_kernel<128, 128>)); else GEMM_LAUNCH((gemm_nt_kernel<64, 64>));
  }
#undef GEMM_LAUNCH
  SAM6D_LAUNCH_CHECK("gemm_nt");
}

extern "C" int sam6d_gemm_nt(const float* A, const float* W, const float* bias, const float* colscale,
                             const float* residual, float* C, int M, int N, int K, long lda, long ldw, long ldc, long ldr,
                             int batch, long sA, long sW, long sC, long sR, float divisor, int act, void* stream) {
  return gemm_launch(A, W, bias, colscale, residual, C, M, N, K, lda, ldw, ldc, ldr, batch, sA, sW, sC, sR,
                     Batch2{1, 0, 0, 0, 0, 0}, divisor, act, stream);
}

extern "C" int sam6d_gemm_nt_w16(const float* A, const float* W, const void* Wh, const void* Wl, float w_scale, const float* bias,
                                 const float* colscale, const float* residual, float* C, int M, int N, int K, long lda, long ldw,
                                 long ldc, long ldr, int batch, long sA, long sW, long sC, long sR, float divisor, int act,
                                 void* stream) {
  SAM6D_REQUIRE(Wh && Wl && w_scale > 0.f, "gemm_nt_w16: the pre-split weight halves and their scale are required");
  return gemm_launch(A, W, bias, colscale, residual, C, M, N, K, lda, ldw, ldc, ldr, batch, sA, sW, sC, sR,
                     Batch2{1, 0, 0, 0, 0, 0}, divisor, act, stream, Wh, Wl, w_scale);
}

extern "C" int sam6d_gemm_nt_b2(const float* A, const float* W, float* C, int M, int N, int K, long lda, long ldw, long ldc,
                                int batch, long sA, long sW, long sC, int batch2, long sA2, long sW2, long sC2, void* stream) {
  return gemm_launch(A, W, nullptr, nullptr, nullptr, C, M, N, K, lda, ldw, ldc, 0, batch, sA, sW, sC, 0,
                     Batch2{batch2, sA2, sW2, sC2, 0, 0}, 1.0f, 0, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM + bias + residual + LayerNorm in one kernel for the 256-wide projections that every transformer block ends with
// (attention.linear -> +x -> norm, output.squeeze -> +y -> norm: PEM/model/transformer.py:152-199, 436-479, 597-622):
//     Y[m, :] = LayerNorm(A[m, :] . W^T + bias + R[m, :]) * gamma + beta          N = 256 fixed
// One workgroup owns 64 full rows (BN = 256: wave w has columns [64 w, 64 w + 64)), so the row statistics never leave the
// chip: per-lane partial sums -> DPP / permlane reduction over the 32 column lanes -> 4 wave partials through LDS -> mean, then
// the same for the centred squares (two-pass variance like layernorm256_kernel).  A is read once (the 128 x 128 tiling reads it
// once per column tile) and the separate LayerNorm launch with its 2 x M x 1 KiB of traffic disappears.
// Split-precision (fp16 x3) arithmetic as gemm_nt_h3_kernel; staging identical.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gemm_ln_h3_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias, const float* __restrict__ residual,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ Y, int M, int K, long lda, long ldw, long ldr,
                                                         long ldy, float eps) {
  constexpr int BM = 64, BN = 256, RA = 2, RB = 8;
  __shared__ __attribute__((aligned(16))) _Float16 smem[2 * (BM + BN) * H_LD];
  __shared__ float red[4][BM];
  __shared__ float stat[BM];
  _Float16* Ah = smem;
  _Float16* Al = Ah + BM * H_LD;
  _Float16* Bh = Al + BM * H_LD;
  _Float16* Bl = Bh + BN * H_LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int m0 = blockIdx.x * BM, wn = wave * 64;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sr = t >> 3, sk = (t & 7) * 4;
  const float* ap[RA];
  const float* bp[RB];
#pragma unroll
  for (int u = 0; u < RA; ++u) ap[u] = A + (size_t)min(m0 + sr + 32 * u, M - 1) * lda + sk;
#pragma unroll
  for (int u = 0; u < RB; ++u) bp[u] = W + (size_t)(sr + 32 * u) * ldw + sk;
  float4 va[RA], vb[RB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < RA; ++u) va[u] = *reinterpret_cast<const float4*>(ap[u] + k0);
#pragma unroll
    for (int u = 0; u < RB; ++u) vb[u] = *reinterpret_cast<const float4*>(bp[u] + k0);
  };
  const int fr = lane & 31, fk = lane >> 5;
  fetch(0);
  for (int k0 = 0; k0 < K; k0 += H_BK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RA; ++u) {
      half4 hi, lo;
      split4(va[u], hi, lo);
      *reinterpret_cast<half4*>(&Ah[(sr + 32 * u) * H_LD + sk]) = hi;
      *reinterpret_cast<half4*>(&Al[(sr + 32 * u) * H_LD + sk]) = lo;
    }
#pragma unroll
    for (int u = 0; u < RB; ++u) {
      half4 hi, lo;
      split4(vb[u], hi, lo);
      *reinterpret_cast<half4*>(&Bh[(sr + 32 * u) * H_LD + sk]) = hi;
      *reinterpret_cast<half4*>(&Bl[(sr + 32 * u) * H_LD + sk]) = lo;
    }
    __syncthreads();
    if (k0 + H_BK < K) fetch(k0 + H_BK);
#pragma unroll
    for (int ks = 0; ks < H_BK; ks += 16) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(&Ah[(32 * i + fr) * H_LD + ks + 8 * fk]);
        al[i] = *reinterpret_cast<const half8*>(&Al[(32 * i + fr) * H_LD + ks + 8 * fk]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        bh[j] = *reinterpret_cast<const half8*>(&Bh[(wn + 32 * j + fr) * H_LD + ks + 8 * fk]);
        bl[j] = *reinterpret_cast<const half8*>(&Bl[(wn + 32 * j + fr) * H_LD + ks + 8 * fk]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // ---- epilogue: x = acc + bias + residual (kept in the accumulator registers)
  float bv[2], gv[2], tv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn + 32 * j + fr;
    bv[j] = bias ? bias[col] : 0.f;
    gv[j] = gamma[col];
    tv[j] = beta[col];
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float rv[2][16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
        rv[j][r] = (residual && row < M) ? residual[(size_t)row * ldr + wn + 32 * j + fr] : 0.f;
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] + bv[j]) + rv[j][r];
  }
  // row sums over this wave's 64 columns: 2 per lane, then the 32 lanes that share fk (row16 all-reduce + the lane 16 away)
  auto wave_rows = [&](auto f) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float p = f(i, r);
        p = row16_sum_dpp(p);
        p += xor16_f32(p);
        if (fr == 0) red[wave][32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk] = p;
      }
  };
  wave_rows([&](int i, int r) { return acc[i][0][r] + acc[i][1][r]; });
  __syncthreads();
  if (t < BM) stat[t] = ((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) * (1.0f / 256.0f);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float mean = stat[32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk];
      acc[i][0][r] -= mean;
      acc[i][1][r] -= mean;
    }
  __syncthreads();  // every lane has read `stat` before it is rewritten with the variances
  wave_rows([&](int i, int r) { return acc[i][0][r] * acc[i][0][r] + acc[i][1][r] * acc[i][1][r]; });
  __syncthreads();
  if (t < BM) stat[t] = 1.0f / sqrtf(((red[0][t] + red[1][t]) + (red[2][t] + red[3][t])) * (1.0f / 256.0f) + eps);
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int rl = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk;
      const int row = m0 + rl;
      if (row < M) {
        const float rstd = stat[rl];
#pragma unroll
        for (int j = 0; j < 2; ++j) Y[(size_t)row * ldy + wn + 32 * j + fr] = acc[i][j][r] * rstd * gv[j] + tv[j];
      }
    }
}

extern "C" int sam6d_gemm_ln256(const float* A, const float* W, const float* bias, const float* residual, const float* gamma,
                                const float* beta, float* Y, int M, int K, long lda, long ldw, long ldr, long ldy, float eps,
                                void* stream) {
  SAM6D_REQUIRE(A && W && gamma && beta && Y, "gemm_ln256: null pointer");
  SAM6D_REQUIRE(M >= 0 && K >= 32 && (K % 32) == 0, "gemm_ln256: K must be a positive multiple of 32 (got %d)", K);
  SAM6D_REQUIRE(lda >= K && ldw >= K && ldy >= 256 && (!residual || ldr >= 256), "gemm_ln256: leading dimension too small");
  SAM6D_REQUIRE(((lda | ldw) & 3) == 0 && ((((size_t)A) | ((size_t)W)) & 15) == 0, "gemm_ln256: A and W rows must be 16-byte aligned");
  SAM6D_REQUIRE(sam6d_get_matmul_mode() >= 1, "gemm_ln256: split-precision mode only (use sam6d_gemm_nt + sam6d_layernorm256 in mode 0)");
  if (M == 0) return 0;
  hipLaunchKernelGGL(gemm_ln_h3_kernel, dim3((unsigned)cdiv(M, 64)), dim3(256), 0, (hipStream_t)stream, A, W, bias, residual, gamma,
                     beta, Y, M, K, lda, ldw, ldr, ldy, eps);
  SAM6D_LAUNCH_CHECK("gemm_ln256");
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm over rows of C=256 (nn.LayerNorm, eps 1e-5; PEM/model/transformer.py:158,189,436,597): one wave per
// row, 4 floats per lane, two-pass mean/variance in registers.  The residual add is fused into the producing GEMM.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm256_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ b, float* __restrict__ y, long rows,
                                                           long ldx, long ldy, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + lane * 4);
  const float mean = wave_sum_dpp((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum_dpp((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f);
  const float rstd = 1.0f / sqrtf(var + eps);
  const float4 gg = *reinterpret_cast<const float4*>(g + lane * 4);
  const float4 bb = *reinterpret_cast<const float4*>(b + lane * 4);
  float4 o;
  o.x = dx * rstd * gg.x + bb.x;
  o.y = dy * rstd * gg.y + bb.y;
  o.z = dz * rstd * gg.z + bb.z;
  o.w = dw * rstd * gg.w + bb.w;
  *reinterpret_cast<float4*>(y + row * ldy + lane * 4) = o;
}

extern "C" int sam6d_layernorm256(const float* x, const float* gamma, const float* beta, float* y, long rows, long ldx,
                                  long ldy, float eps, void* stream) {
  SAM6D_REQUIRE(x && gamma && beta && y, "layernorm256: null pointer");
  SAM6D_REQUIRE(rows >= 0 && ldx >= 256 && ldy >= 256 && (ldx & 3) == 0 && (ldy & 3) == 0, "layernorm256: bad sizes");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(layernorm256_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, gamma,
                     beta, y, rows, ldx, ldy, eps);
  SAM6D_LAUNCH_CHECK("layernorm256");
}
