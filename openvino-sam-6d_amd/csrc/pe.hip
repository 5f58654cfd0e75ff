// The fine-matching PositionalEncoding (PEM/model/fine_point_matching.py:102-144) as one fused matrix-core kernel per
// scale, and small row utilities of the dense path.
#include "common.h"
#include "../../include/sam6d_hip.h"

// ===============================================================================================================
// Fused QueryAndGroup -> SharedMLP(6->32->64->128, Conv2d 1x1 + BatchNorm(eval) + ReLU) -> max over the ball
// (PEM/model/fine_point_matching.py:126-139; PEM/model/pointnet2/pointnet2_utils.py:383-396; pytorch_utils.py:25-50).
//
// One wave owns a query point: its S neighbours are the 32 rows of an MFMA tile (S = 64: two tiles), so the three
// layers are chained v_mfma_f32_32x32x2_f32 products (3 + 32 + 128 per tile, exact fp32) whose outputs go
// BN/ReLU -> a private LDS slab -> next layer's A operand; the 128 channel maxima are reduced in registers and only
// 512 B per point and scale reach HBM (the unfused form wrote and re-read 230 floats per neighbour: 5.6 GB per scale at
// B = 32).  Weights + BN constants of the scale live in LDS (44 KB), 8 waves per workgroup.
// Layer-1 features are formed in registers: {p_idx - (p_j + 1e-8), p_idx}  (new_xyz = pts + 1e-8, :117).
// ===============================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define PM_WAVES 8
#define PM_PPW 8
#define PM_WFLOATS (32 * 7 + 64 * 33 + 128 * 65 + 448)
#define PM_HFLOATS (32 * 33 + 32 * 65)

__global__ __launch_bounds__(PM_WAVES * 64) void pe_mlp_max_kernel(
    const float* __restrict__ pts, const int* __restrict__ idx, int N, int S, long total, const float* __restrict__ W1,
    const float* __restrict__ sc1, const float* __restrict__ sh1, const float* __restrict__ W2, const float* __restrict__ sc2,
    const float* __restrict__ sh2, const float* __restrict__ W3, const float* __restrict__ sc3, const float* __restrict__ sh3,
    float* __restrict__ out, long ldo, int off) {
  extern __shared__ float lds[];
  float* w1s = lds;             // [32][7]
  float* w2s = w1s + 32 * 7;    // [64][33]
  float* w3s = w2s + 64 * 33;   // [128][65]
  float* bn = w3s + 128 * 65;   // sc1 32 | sh1 32 | sc2 64 | sh2 64 | sc3 128 | sh3 128
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float* h1 = bn + 448 + wave * PM_HFLOATS;  // [32][33]
  float* h2 = h1 + 32 * 33;                  // [32][65]
  for (int e = t; e < 32 * 6; e += PM_WAVES * 64) w1s[(e / 6) * 7 + (e % 6)] = W1[e];
  for (int e = t; e < 64 * 32; e += PM_WAVES * 64) w2s[(e >> 5) * 33 + (e & 31)] = W2[e];
  for (int e = t; e < 128 * 64; e += PM_WAVES * 64) w3s[(e >> 6) * 65 + (e & 63)] = W3[e];
  if (t < 32) { bn[t] = sc1[t]; bn[32 + t] = sh1[t]; }
  if (t < 64) { bn[64 + t] = sc2[t]; bn[128 + t] = sh2[t]; }
  if (t < 128) { bn[192 + t] = sc3[t]; bn[320 + t] = sh3[t]; }
  __syncthreads();
  const int fr = lane & 31, fk = lane >> 5;
  const int ntile = S >> 5;
  for (int i = 0; i < PM_PPW; ++i) {
    const long p = ((long)blockIdx.x * PM_WAVES + wave) * PM_PPW + i;  // flat point id over B*N
    if (p >= total) break;
    const long b = p / N;
    const float* pb = pts + b * N * 3;
    const float qx = pts[p * 3] + 0.00000001f, qy = pts[p * 3 + 1] + 0.00000001f, qz = pts[p * 3 + 2] + 0.00000001f;
    float mx[4] = {0.f, 0.f, 0.f, 0.f};  // ReLU outputs are >= 0, so 0 is a neutral start for the max
    for (int tile = 0; tile < ntile; ++tile) {
      const int nb = idx[p * S + tile * 32 + fr];
      const bool ok = nb >= 0 && nb < N;
      const float x = ok ? pb[nb * 3] : 0.f, y = ok ? pb[nb * 3 + 1] : 0.f, z = ok ? pb[nb * 3 + 2] : 0.f;
      // ---- layer 1: K = 6 (k = 2s + fk): {x-qx, y-qy, z-qz, x, y, z}
      const float f0 = fk ? (y - qy) : (x - qx);
      const float f1 = fk ? x : (z - qz);
      const float f2 = fk ? z : y;
      f32x16 a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) a1[r] = 0.f;
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f0, w1s[fr * 7 + 0 + fk], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f1, w1s[fr * 7 + 2 + fk], a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(f2, w1s[fr * 7 + 4 + fk], a1, 0, 0, 0);
      {
        const float s = bn[fr], h = bn[32 + fr];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaf(a1[r], s, h);
          h1[((r & 3) + 8 * (r >> 2) + 4 * fk) * 33 + fr] = v > 0.f ? v : 0.f;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- layer 2: K = 32, 2 column tiles
      f32x16 a2[2];
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[c][r] = 0.f;
#pragma unroll 4
      for (int s = 0; s < 16; ++s) {
        const float a = h1[fr * 33 + 2 * s + fk];
        a2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w2s[fr * 33 + 2 * s + fk], a2[0], 0, 0, 0);
        a2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w2s[(32 + fr) * 33 + 2 * s + fk], a2[1], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const float s = bn[64 + c * 32 + fr], h = bn[128 + c * 32 + fr];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = fmaf(a2[c][r], s, h);
          h2[((r & 3) + 8 * (r >> 2) + 4 * fk) * 65 + c * 32 + fr] = v > 0.f ? v : 0.f;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- layer 3: K = 64, 4 column tiles
      f32x16 a3[4];
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) a3[c][r] = 0.f;
#pragma unroll 4
      for (int s = 0; s < 32; ++s) {
        const float a = h2[fr * 65 + 2 * s + fk];
#pragma unroll
        for (int c = 0; c < 4; ++c)
          a3[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w3s[(c * 32 + fr) * 65 + 2 * s + fk], a3[c], 0, 0, 0);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float s = bn[192 + c * 32 + fr], h = bn[320 + c * 32 + fr];
        float m = mx[c];
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fmaf(a3[c][r], s, h));
        mx[c] = m;
      }
      __builtin_amdgcn_wave_barrier();  // h1/h2 are rewritten by the next tile only after every lane has read them
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float m = fmaxf(mx[c], __shfl_xor(mx[c], 32, 64));
      if (fk == 0) out[p * ldo + off + c * 32 + fr] = m;
    }
  }
}

// ===============================================================================================================
// fp16 x3 split-precision form (default matmul mode), register-chained: activations never leave the registers.
// Layer 1 (K = 6, exact fp32 MFMA) and layer 2 are computed TRANSPOSED (weights as the A operand, neighbours as the
// columns), so each lane ends a layer holding, for its own neighbour, 16 channels per 32x32 tile -- which is exactly
// the shape of the next product's K-operand (8 consecutive k per lane half) up to a fixed permutation of k: channel
// (j&3) + 8*(2s + (j>>2)) + 4*fk sits at element j of k-step s, i.e. k with bits 2 and 3 swapped.  The weights are
// staged into LDS with that permutation, so BN/ReLU -> fp16 hi/lo split -> pack is all that separates two layers (no
// LDS slab, no wave barrier, no 16-bit stores).  Layer 3 consumes the layer-2 registers as its A operand (rows =
// neighbours), so the max over the ball stays a register max per output channel as before.
// 12 + 48 v_mfma_f32_32x32x16_f16 + 3 v_mfma_f32_32x32x2_f32 per 32-neighbour tile (2112 matrix-pipe cycles).
// Persistent waves (3 workgroups of 4 waves per CU) stride over the points; the neighbour indices are loaded two
// tiles ahead and the gathered coordinates one tile ahead, so the idx -> xyz dependent loads hide behind a tile's MFMAs.
// ===============================================================================================================
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define PH_WAVES 4
#define PH_L2 40   // halves per row of the K=32 images (80 B)
#define PH_L3 72   // halves per row of the K=64 images (144 B = 9 x 16 B: 16 consecutive rows hit 16 different slots)
#define PH_WBYTES (32 * 7 * 4 + 448 * 4 + 2 * 64 * PH_L2 * 2 + 2 * 128 * PH_L3 * 2)

__device__ __forceinline__ int pe_swap23(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

// BN(eval) + ReLU + fp16 hi/lo split of the 8 accumulator registers [8*half, 8*half + 8) of one transposed tile; the
// channels of those registers are base + 8*g + 4*fk + e (g = 2*half + (j>>2), e = j&3): two 16-byte constant reads each.
// PRE: the accumulators are first multiplied by `pre` (layer 1: the inverse of this lane's feature scale times the inverse W1 scale).
__device__ __forceinline__ float pe_pow2_scale(float amax) {
  // power of two s with amax * s in [2^13, 2^14); 1 for zero / non-finite input
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  e = 14 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}

template <bool PRE = false>
__device__ __forceinline__ void pe_split8(const f32x16& acc, int half, const float* __restrict__ sc, const float* __restrict__ sh,
                                          int fk, half8& hi, half8& lo, float pre = 1.0f) {
#pragma unroll
  for (int g2 = 0; g2 < 2; ++g2) {
    const int g = 2 * half + g2;
    const f32x4 s4 = *reinterpret_cast<const f32x4*>(&sc[8 * g + 4 * fk]);
    const f32x4 h4 = *reinterpret_cast<const f32x4*>(&sh[8 * g + 4 * fk]);
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
      float a0 = acc[4 * g + e], a1 = acc[4 * g + e + 1];
      if constexpr (PRE) { a0 *= pre; a1 *= pre; }
      float v0 = fmaf(a0, s4[e], h4[e]), v1 = fmaf(a1, s4[e + 1], h4[e + 1]);
      v0 = v0 > 0.f ? v0 : 0.f;
      v1 = v1 > 0.f ? v1 : 0.f;
      unsigned ph, pl;
      sam6d_split2_f16(v0, v1, ph, pl);
      const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), ph);
      const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), pl);
      hi[4 * g2 + e] = h2[0];
      hi[4 * g2 + e + 1] = h2[1];
      lo[4 * g2 + e] = l2[0];
      lo[4 * g2 + e + 1] = l2[1];
    }
  }
}

__global__ __launch_bounds__(PH_WAVES * 64) __attribute__((amdgpu_waves_per_eu(3, 3))) void pe_mlp_max_h3_kernel(
    const float* __restrict__ pts, const int* __restrict__ idx, int N, int S, int total, const float* __restrict__ W1,
    const float* __restrict__ sc1, const float* __restrict__ sh1, const float* __restrict__ W2, const float* __restrict__ sc2,
    const float* __restrict__ sh2, const float* __restrict__ W3, const float* __restrict__ sc3, const float* __restrict__ sh3,
    float* __restrict__ out, long ldo, int off, unsigned ndiv) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lraw[];
  float* w1s = reinterpret_cast<float*>(lraw);                  // [32][7] fp32
  float* bn = w1s + 32 * 7;                                     // sc1 32 | sh1 32 | sc2 64 | sh2 64 | sc3 128 | sh3 128
  _Float16* w2h = reinterpret_cast<_Float16*>(bn + 448);        // [64][40], k permuted
  _Float16* w2l = w2h + 64 * PH_L2;
  _Float16* w3h = w2l + 64 * PH_L2;                             // [128][72], k permuted
  _Float16* w3l = w3h + 128 * PH_L3;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // W1 as it is staged: times the power of two that puts max |W1| into [2^13, 2^14) (its fp16 lo halves then stay normal numbers); the
  // inverse goes into the per-lane unscale of the layer-1 accumulators below.  Every workgroup computes the same scale.
  float w1max = 0.f;
  for (int e = t; e < 32 * 6; e += PH_WAVES * 64) w1max = fmaxf(w1max, fabsf(W1[e]));
  w1max = wave_max_dpp(w1max);
  if (lane == 0) w1s[7 * wave + 6] = w1max;  // (column 6 of the [32][7] image is padding)
  __syncthreads();
  {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < PH_WAVES; ++w) m = fmaxf(m, w1s[7 * w + 6]);
    w1max = m;
  }
  __syncthreads();
  const float w1scale = pe_pow2_scale(w1max), w1inv = 1.0f / w1scale;
  for (int e = t; e < 32 * 6; e += PH_WAVES * 64) w1s[(e / 6) * 7 + (e % 6)] = W1[e] * w1scale;
  for (int e = t; e < 64 * 32; e += PH_WAVES * 64) {
    const float v = W2[e];
    const _Float16 h = (_Float16)v;
    const int o = (e >> 5) * PH_L2 + pe_swap23(e & 31);
    w2h[o] = h;
    w2l[o] = (_Float16)(v - (float)h);
  }
  // Layer 3 ends in max over the ball of BN(acc) = fma(acc, sc3, sh3).  With the SIGN of sc3[ch] folded into row ch of W3 (exact) the
  // map acc -> fma(acc, |sc3|, sh3) is monotone non-decreasing, so the max commutes with it bit for bit: the per-tile epilogue is a
  // running max over the raw accumulators (v_max3: 8 instructions per 16 values instead of 16 fma + 16 max) and the fma runs once per
  // point (stamps of round 3: the max epilogue cost 1.4-1.7 k cycles per tile beside 2 k cycles of MFMA issue).
  for (int e = t; e < 128 * 64; e += PH_WAVES * 64) {
    const float v = sc3[e >> 6] < 0.f ? -W3[e] : W3[e];
    const _Float16 h = (_Float16)v;
    const int o = (e >> 6) * PH_L3 + pe_swap23(e & 63);
    w3h[o] = h;
    w3l[o] = (_Float16)(v - (float)h);
  }
  if (t < 32) { bn[t] = sc1[t]; bn[32 + t] = sh1[t]; }
  if (t < 64) { bn[64 + t] = sc2[t]; bn[128 + t] = sh2[t]; }
  if (t < 128) { bn[192 + t] = fabsf(sc3[t]); bn[320 + t] = sh3[t]; }
  __syncthreads();
  const int fr = lane & 31, fk = lane >> 5;
  const int ntile = S >> 5;
  // cloud of point p: p / N by a multiply-high with ndiv = floor(2^32 / N) + 1 (exact while p N < 2^32: checked on the host), so the
  // prefetch code of a tile is straight-line -- the loads below are unconditional on clamped indices, the selects follow (stamps: the
  // branchy form with its integer division cost each wave ~1 350 cycles per tile)
  auto cloud_base = [&](int p) -> long { return (long)(ndiv ? (int)__umulhi((unsigned)p, ndiv) : p / N) * N * 3; };
  const int GW = gridDim.x * PH_WAVES;
  float s3[4], h3[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) { s3[c] = bn[192 + c * 32 + fr]; h3[c] = bn[320 + c * 32 + fr]; }
  // tile A = being computed, B = next (indices loaded, coordinates in flight), C = the one after (indices in flight)
  int pA = __builtin_amdgcn_readfirstlane(blockIdx.x * PH_WAVES + wave), tA = 0;
  int pB = pA, tB = 1;
  if (tB == ntile) { tB = 0; pB += GW; }
  int pC = pB, tC = tB + 1;
  if (tC == ntile) { tC = 0; pC += GW; }
  float xA = 0.f, yA = 0.f, zA = 0.f, qxA = 0.f, qyA = 0.f, qzA = 0.f;
  int nbB = 0;
  if (pA < total) {
    const int nb = idx[(long)pA * S + tA * 32 + fr];
    const float* pb = pts + cloud_base(pA);
    const bool ok = nb >= 0 && nb < N;
    xA = ok ? pb[nb * 3] : 0.f; yA = ok ? pb[nb * 3 + 1] : 0.f; zA = ok ? pb[nb * 3 + 2] : 0.f;
    qxA = pts[(long)pA * 3] + 0.00000001f; qyA = pts[(long)pA * 3 + 1] + 0.00000001f; qzA = pts[(long)pA * 3 + 2] + 0.00000001f;
  }
  nbB = idx[(long)min(pB, total - 1) * S + tB * 32 + fr];
  // layer-1 weight fragments of this lane (row = channel fr): w1a = hi halves (lanes 0-31) / lo halves (lanes 32-63) of W1[fr][0..5],
  // w1b = hi halves in lanes 0-31, zeros above
  typedef unsigned pe_u4 __attribute__((ext_vector_type(4)));
  half8 w1a, w1b;
  {
    float wv[6];
#pragma unroll
    for (int f = 0; f < 6; ++f) wv[f] = w1s[fr * 7 + f];
    unsigned h01, l01, h23, l23, h45, l45;
    sam6d_split2_f16(wv[0], wv[1], h01, l01);
    sam6d_split2_f16(wv[2], wv[3], h23, l23);
    sam6d_split2_f16(wv[4], wv[5], h45, l45);
    w1a = __builtin_bit_cast(half8, fk ? pe_u4{l01, l23, l45, 0u} : pe_u4{h01, h23, h45, 0u});
    w1b = __builtin_bit_cast(half8, fk ? pe_u4{0u, 0u, 0u, 0u} : pe_u4{h01, h23, h45, 0u});
  }
  float mx[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};  // running max of the raw layer-3 accumulators of the point
  int wrow = fr, wk = 8 * fk;  // this lane's row / k offset in the weight images
  while (pA < total) {
    // the weight reads are loop-invariant: keep them inside the loop (hoisted, they would not fit the register file)
    asm volatile("" : "+v"(wrow), "+v"(wk));
    // ---- prefetch: indices of tile C, coordinates of tile B
    const int nbC = idx[(long)min(pC, total - 1) * S + tC * 32 + fr];  // (past the end: the last point's indices, never used)
    float xB, yB, zB, qxB, qyB, qzB;
    {
      const int pBc = min(pB, total - 1);
      const float* pb = pts + cloud_base(pBc);
      const bool ok = nbB >= 0 && nbB < N;
      const int nbc = ok ? nbB : 0;
      xB = pb[nbc * 3]; yB = pb[nbc * 3 + 1]; zB = pb[nbc * 3 + 2];
      xB = ok ? xB : 0.f; yB = ok ? yB : 0.f; zB = ok ? zB : 0.f;
      qxB = pts[(long)pBc * 3] + 0.00000001f; qyB = pts[(long)pBc * 3 + 1] + 0.00000001f; qzB = pts[(long)pBc * 3 + 2] + 0.00000001f;
    }
    // ---- layer 1 (transposed): D1T[ch][nb] = W1[ch][f] F[f][nb], f: {x-qx, y-qy, z-qz, x, y, z}.  K = 6 leaves room for all three
    // split products in TWO v_mfma_f32_32x32x16_f16: k slots 0..5 of the lower lane half carry w_hi . x_hi, of the upper half w_lo . x_hi
    // (first instruction), then w_hi . x_lo in the lower half (second).  (Was three v_mfma_f32_32x32x2_f32 = 192 cycles per tile during
    // which the SIMD's other waves cannot issue vector instructions; these are 64 cycles and overlap with them.)
    f32x16 a1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a1[r] = 0.f;
    // Range safety of the split (round 4): the six features of this lane's neighbour are scaled by the power of two that puts their
    // largest magnitude into [2^13, 2^14) -- raw coordinates below 0.125 had fp16-subnormal lo halves (an ABSOLUTE error floor of 3e-8
    // instead of a relative one) and coordinates >= 65504 overflowed to inf, where the reference is fine.  The scale is per lane
    // (= per neighbour = per column of the transposed product), so it is undone exactly on the lane's own accumulators.
    float finv;
    {
      const float f0 = xA - qxA, f1 = yA - qyA, f2 = zA - qzA;
      const float fm = fmaxf(fmaxf(fmaxf(fabsf(f0), fabsf(f1)), fmaxf(fabsf(f2), fabsf(xA))), fmaxf(fabsf(yA), fabsf(zA)));
      const int ex = (int)(__float_as_uint(fm) >> 23);  // biased exponent: fm in [2^(ex-127), 2^(ex-126))
      const int se = min(267 - ex, 240);                // biased exponent of the scale 2^(140 - ex), capped for fm ~ 0
      const float fs = __uint_as_float((unsigned)se << 23);
      finv = __uint_as_float((unsigned)(254 - se) << 23) * w1inv;
      unsigned h01, l01, h23, l23, h45, l45;
      sam6d_split2_f16(f0 * fs, f1 * fs, h01, l01);
      sam6d_split2_f16(f2 * fs, xA * fs, h23, l23);
      sam6d_split2_f16(yA * fs, zA * fs, h45, l45);
      const half8 bx = __builtin_bit_cast(half8, pe_u4{h01, h23, h45, 0u});
      const half8 bl = __builtin_bit_cast(half8, fk ? pe_u4{0u, 0u, 0u, 0u} : pe_u4{l01, l23, l45, 0u});
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1a, bx, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1b, bl, a1, 0, 0, 0);
    }
    half8 h1h[2], h1l[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) pe_split8<true>(a1, s, bn, bn + 32, wk >> 3, h1h[s], h1l[s], finv);
    // ---- layer 2 (transposed): D2T[ch2][nb] = W2[ch2][k] H1T[k][nb]
    f32x16 a2[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) a2[c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const half8 wh = *reinterpret_cast<const half8*>(&w2h[(c * 32 + wrow) * PH_L2 + 16 * s + wk]);
        const half8 wl = *reinterpret_cast<const half8*>(&w2l[(c * 32 + wrow) * PH_L2 + 16 * s + wk]);
        a2[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, h1l[s], a2[c], 0, 0, 0);
        a2[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, h1h[s], a2[c], 0, 0, 0);
        a2[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, h1h[s], a2[c], 0, 0, 0);
      }
    }
    // ---- layer 3: D3[nb][ch3] = H2[nb][k] W3[ch3][k]^T, k-step s = registers [8(s&1), +8) of tile s>>1
    f32x16 a3[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 16; ++r) a3[c][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      half8 ah, al;
      pe_split8(a2[s >> 1], s & 1, bn + 64 + 32 * (s >> 1), bn + 128 + 32 * (s >> 1), wk >> 3, ah, al);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const half8 bh = *reinterpret_cast<const half8*>(&w3h[(c * 32 + wrow) * PH_L3 + 16 * s + wk]);
        const half8 bl = *reinterpret_cast<const half8*>(&w3l[(c * 32 + wrow) * PH_L3 + 16 * s + wk]);
        a3[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, a3[c], 0, 0, 0);
        a3[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, a3[c], 0, 0, 0);
        a3[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, a3[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      float m = mx[c];
#pragma unroll
      for (int r = 0; r < 16; r += 2) m = fmaxf(fmaxf(m, a3[c][r]), a3[c][r + 1]);
      mx[c] = m;
    }
    if (tA == ntile - 1) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float m = fmaxf(mx[c], xor32_f32(mx[c]));
        m = fmaxf(fmaf(m, s3[c], h3[c]), 0.f);  // BN (|scale|: the sign is in W3) + ReLU of the maximum = the maximum of BN + ReLU
        if (fk == 0) out[(long)pA * ldo + off + c * 32 + fr] = m;
        mx[c] = -INFINITY;
      }
    }
    // ---- rotate the pipeline
    pA = pB; tA = tB; pB = pC; tB = tC;
    if (++tC == ntile) { tC = 0; pC += GW; }
    nbB = nbC;
    xA = xB; yA = yB; zA = zB; qxA = qxB; qyA = qyB; qzA = qzB;
  }
}

// max_wg: upper bound on the persistent workgroups of the split-precision kernel (0 = fill the chip: 768 = 3 per CU).  A launch that
// shares the chip with another stream's kernels leaves room with 256 or 512 (1 or 2 per CU: 49 / 97 KB of LDS, 150 / 300 VGPRs per SIMD).
static int pe_mlp_max_impl(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1, const float* sh1,
                           const float* W2, const float* sc2, const float* sh2, const float* W3, const float* sc3, const float* sh3,
                           float* out, long ldo, int off, int max_wg, void* stream);
extern "C" int sam6d_pe_mlp_max(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1,
                                const float* sh1, const float* W2, const float* sc2, const float* sh2, const float* W3,
                                const float* sc3, const float* sh3, float* out, long ldo, int off, void* stream) {
  return pe_mlp_max_impl(pts, idx, B, N, S, W1, sc1, sh1, W2, sc2, sh2, W3, sc3, sh3, out, ldo, off, 0, stream);
}
extern "C" int sam6d_pe_mlp_max_wg(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1,
                                   const float* sh1, const float* W2, const float* sc2, const float* sh2, const float* W3,
                                   const float* sc3, const float* sh3, float* out, long ldo, int off, int max_wg, void* stream) {
  SAM6D_REQUIRE(max_wg >= 0, "pe_mlp_max_wg: max_wg must be >= 0");
  return pe_mlp_max_impl(pts, idx, B, N, S, W1, sc1, sh1, W2, sc2, sh2, W3, sc3, sh3, out, ldo, off, max_wg, stream);
}
static int pe_mlp_max_impl(const float* pts, const int* idx, int B, int N, int S, const float* W1, const float* sc1, const float* sh1,
                           const float* W2, const float* sc2, const float* sh2, const float* W3, const float* sc3, const float* sh3,
                           float* out, long ldo, int off, int max_wg, void* stream) {
  SAM6D_REQUIRE(pts && idx && W1 && sc1 && sh1 && W2 && sc2 && sh2 && W3 && sc3 && sh3 && out, "pe_mlp_max: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && S > 0 && (S & 31) == 0, "pe_mlp_max: nsample must be a multiple of 32 (got %d)", S);
  const long total = (long)B * N;
  if (total == 0) return 0;
  const long per_block = PM_WAVES * PM_PPW;
  const dim3 grid((unsigned)((total + per_block - 1) / per_block));
  if (sam6d_get_matmul_mode() >= 1) {
    SAM6D_REQUIRE(total < (1l << 31) / 64, "pe_mlp_max: B*N too large for 32-bit point ids (%ld)", total);
    const size_t lds = (size_t)PH_WBYTES;
    static unsigned long long attr_h3 = 0;
    if (sam6d_first_use_on_device(&attr_h3)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pe_mlp_max_h3_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        sam6d_set_error("pe_mlp_max: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      sam6d_setup_done_on_device(&attr_h3);
    }
    // persistent waves: 3 workgroups of 4 waves fit one CU's LDS (3 x 48.6 KB) -> 768 workgroups fill the 256 CUs once
    const long want = (total + PH_WAVES - 1) / PH_WAVES;
    const long cap = (max_wg > 0 && max_wg < 768) ? max_wg : 768;
    const dim3 pgrid((unsigned)(want < cap ? want : cap));
    // p / N inside the kernel by multiply-high: exact while p N < 2^32 (else 0: the kernel divides)
    const unsigned ndiv = (total * (long)N < (1l << 32) && N > 1) ? (unsigned)((1ull << 32) / (unsigned long long)N) + 1u : 0u;
    hipLaunchKernelGGL(pe_mlp_max_h3_kernel, pgrid, dim3(PH_WAVES * 64), lds, (hipStream_t)stream, pts, idx, N, S, (int)total,
                       W1, sc1, sh1, W2, sc2, sh2, W3, sc3, sh3, out, ldo, off, ndiv);
  } else {
    const size_t lds = (size_t)(PM_WFLOATS + PM_WAVES * PM_HFLOATS) * 4;
    static unsigned long long attr_set = 0;
    if (sam6d_first_use_on_device(&attr_set)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pe_mlp_max_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) {
        sam6d_set_error("pe_mlp_max: cannot reserve %zu bytes of LDS: %s", lds, hipGetErrorString(e));
        return (int)e;
      }
      sam6d_setup_done_on_device(&attr_set);
    }
    hipLaunchKernelGGL(pe_mlp_max_kernel, grid, dim3(PM_WAVES * 64), lds, (hipStream_t)stream, pts, idx, N, S, total, W1, sc1,
                       sh1, W2, sc2, sh2, W3, sc3, sh3, out, ldo, off);
  }
  SAM6D_LAUNCH_CHECK("pe_mlp_max");
}

// y[b, i, :] = (x[b, i, :] - t[b]) @ R[b]   (row vector times R; PEM/model/fine_point_matching.py:45, model_utils.py:262,332)
__global__ __launch_bounds__(256) void rigid_inverse_kernel(const float* __restrict__ x, const float* __restrict__ R,
                                                            const float* __restrict__ t, int N, long total,
                                                            float* __restrict__ y) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / N;
  const float* Rb = R + b * 9;
  const float d0 = x[e * 3] - t[b * 3], d1 = x[e * 3 + 1] - t[b * 3 + 1], d2 = x[e * 3 + 2] - t[b * 3 + 2];
  // torch matmul K=3 recipe: fma(d2,R2j, fma(d1,R1j, d0*R0j))
  y[e * 3 + 0] = fmaf(d2, Rb[6], fmaf(d1, Rb[3], d0 * Rb[0]));
  y[e * 3 + 1] = fmaf(d2, Rb[7], fmaf(d1, Rb[4], d0 * Rb[1]));
  y[e * 3 + 2] = fmaf(d2, Rb[8], fmaf(d1, Rb[5], d0 * Rb[2]));
}

extern "C" int sam6d_rigid_inverse(const float* x, const float* R, const float* t, int B, int N, float* y, void* stream) {
  SAM6D_REQUIRE(x && R && t && y && B >= 0 && N > 0, "rigid_inverse: bad arguments");
  const long total = (long)B * N;
  if (total == 0) return 0;
  hipLaunchKernelGGL(rigid_inverse_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, R, t,
                     N, total, y);
  SAM6D_LAUNCH_CHECK("rigid_inverse");
}

// dst[b, row0 + r, :] = src[r or b*rows + r, :] for rows of C floats: writes the learnt bg token (1 row broadcast over
// the batch) or copies a block of rows into a larger token buffer (torch.cat call sites: coarse_point_matching.py:36-38,
// fine_point_matching.py:48-51, transformer.py:703-705,713).
__global__ __launch_bounds__(256) void put_rows_kernel(const float* __restrict__ src, long s_src_b, long ld_src,
                                                       float* __restrict__ dst, long s_dst_b, long ld_dst, int rows, int C,
                                                       long total) {
  const int c4 = C >> 2;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int c = (int)(e % c4) * 4;
  const long br = e / c4;
  const int r = (int)(br % rows);
  const long b = br / rows;
  *reinterpret_cast<float4*>(dst + b * s_dst_b + (long)r * ld_dst + c) =
      *reinterpret_cast<const float4*>(src + b * s_src_b + (long)r * ld_src + c);
}

extern "C" int sam6d_put_rows(const float* src, long s_src_b, long ld_src, float* dst, long s_dst_b, long ld_dst, int B,
                              int rows, int C, void* stream) {
  SAM6D_REQUIRE(src && dst && B >= 0 && rows >= 0 && C > 0 && (C & 3) == 0, "put_rows: bad arguments");
  SAM6D_REQUIRE(((s_src_b | ld_src | s_dst_b | ld_dst) & 3) == 0, "put_rows: strides must be multiples of 4 floats");
  const long total = (long)B * rows * (C >> 2);
  if (total == 0) return 0;
  hipLaunchKernelGGL(put_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, s_src_b,
                     ld_src, dst, s_dst_b, ld_dst, rows, C, total);
  SAM6D_LAUNCH_CHECK("put_rows");
}

// flag[0] &= (x[b] == x[0] bitwise for every b): detects the `.repeat`ed template tensors of the reference's caller
// (PEM/run_inference_custom_pytorch.py:445-446) so that their pose-independent work can run once.  flag is preset to 1 here.
__global__ __launch_bounds__(256) void batch_rows_equal_kernel(const unsigned* __restrict__ x, int B, long n, int* __restrict__ flag) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const unsigned v = x[i];
  bool same = true;
  for (int b = 1; b < B; ++b) same = same && (x[(size_t)b * n + i] == v);
  if (!same) atomicAnd(flag, 0);
}
__global__ void set_flag_kernel(int* p, int v) { *p = v; }

extern "C" int sam6d_batch_rows_equal(const float* x, int B, long n, int* flag, void* stream) {
  SAM6D_REQUIRE(x && flag && B >= 0 && n >= 0, "batch_rows_equal: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, s, flag, 1);
  if (B > 1 && n > 0)
    hipLaunchKernelGGL(batch_rows_equal_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const unsigned*>(x), B,
                       n, flag);
  SAM6D_LAUNCH_CHECK("batch_rows_equal");
}

// cat([bg_point(100,100,100), sparse points]) for the geometric embedding (PEM/model/pose_estimation_model.py:30-34)
__global__ void prepend_bg_point_kernel(const float* __restrict__ p, int n, long total, float* __restrict__ o) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;  // over B*(n+1)
  if (e >= total) return;
  const long b = e / (n + 1);
  const int r = (int)(e % (n + 1));
  float x = 100.f, y = 100.f, z = 100.f;
  if (r > 0) {
    const float* s = p + (b * n + (r - 1)) * 3;
    x = s[0]; y = s[1]; z = s[2];
  }
  o[e * 3] = x; o[e * 3 + 1] = y; o[e * 3 + 2] = z;
}

extern "C" int sam6d_prepend_bg_point(const float* pts, int B, int n, float* out, void* stream) {
  SAM6D_REQUIRE(pts && out && B >= 0 && n > 0, "prepend_bg_point: bad arguments");
  const long total = (long)B * (n + 1);
  if (total == 0) return 0;
  hipLaunchKernelGGL(prepend_bg_point_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts, n,
                     total, out);
  SAM6D_LAUNCH_CHECK("prepend_bg_point");
}

// F.normalize(x, p=2, dim=-1) on rows of 256 floats: x / max(|x|, 1e-12)   (PEM/utils/model_utils.py:141-142)
__global__ __launch_bounds__(256) void l2norm256_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                        long ldx, long ldy) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 v = *reinterpret_cast<const float4*>(x + row * ldx + lane * 4);
  const float n = fmaxf(sqrtf(wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w))), 1e-12f);
  *reinterpret_cast<float4*>(y + row * ldy + lane * 4) = make_float4(v.x / n, v.y / n, v.z / n, v.w / n);
}

extern "C" int sam6d_l2norm256(const float* x, float* y, long rows, long ldx, long ldy, void* stream) {
  SAM6D_REQUIRE(x && y && rows >= 0 && ldx >= 256 && ldy >= 256 && ((ldx | ldy) & 3) == 0, "l2norm256: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(l2norm256_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, rows, ldx, ldy);
  SAM6D_LAUNCH_CHECK("l2norm256");
}

// y = x + s elementwise (new_xyz = pts + 1e-8, PEM/model/fine_point_matching.py:117)
__global__ void add_scalar_kernel(const float* __restrict__ x, float s, long n, float* __restrict__ y) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < n) y[e] = x[e] + s;
}

extern "C" int sam6d_add_scalar(const float* x, float s, long n, float* y, void* stream) {
  SAM6D_REQUIRE(x && y && n >= 0, "add_scalar: bad arguments");
  if (n == 0) return 0;
  hipLaunchKernelGGL(add_scalar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, s, n, y);
  SAM6D_LAUNCH_CHECK("add_scalar");
}

// device-to-device copy of n floats on the stream (the batch-stacking torch.cat / .repeat call sites)
extern "C" int sam6d_copy_f32(const float* src, float* dst, long n, void* stream) {
  SAM6D_REQUIRE(src && dst && n >= 0, "copy_f32: bad arguments");
  if (n == 0) return 0;
  hipError_t e = hipMemcpyAsync(dst, src, (size_t)n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream);
  if (e != hipSuccess) {
    sam6d_set_error("copy_f32: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}
