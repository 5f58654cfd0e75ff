// RPE self-attention scores without a materialised geometric embedding (gfx950).
//
//   RPETransformerLayer (PEM/model/transformer.py:366-420):  s[h,n,m] = (q_h[n].k_h[m] + q_h[n].proj_p(E[n,m])_h) / sqrt(64)
//   E[n,m,:] = proj_d(sinusoid(d_idx[n,m])) + max_k proj_a(sinusoid(a_idx[n,m,k]))      (transformer.py:343-363)
//
// attention.hip streams E (1 KiB per pair, 2.5 GB per layer at B = 32) from HBM.  Here E is never written: with proj_p
// folded into the query (qp[n,h,:] = W_p,h^T q_h, attention.hip) and proj_d / proj_a replaced by their 32-term Chebyshev
// expansions D_c, A_c on [0, xmax] (geo.hip 3c),
//
//   qp_h . E[n,m] = (D_c^T qp_h) . T(u_d)  +  sum_c qp_h[c] * max_k (A_c T(u_a,k))[c]   (+ terms constant in m, which cancel
//                    \__ qd[n,h,:] (32)                                                    in the softmax: both biases)
//
// the d part is a 32-term dot per (pair, head); the a part is a (16 keys x 3) x 32 x 256 contraction per key tile on
// v_mfma_f32_16x16x32_f16 (fp16 hi/lo split, 3 products), followed by max over k and a second, small MFMA contraction of the
// maxima with the folded query (the 4 head dots).  Pairs outside [0, xmax] (the bg token: 2n-1 of n^2) take their
// bias-free E row from the compact buffer sam6d_geo_outliers filled, old-style (one wave per row).
// q.k comes in precomputed (one small batched GEMM), softmax happens here, P.V is a batched GEMM afterwards.
//
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#define RP_ROW 144                 // bytes per coefficient row image: 32 hi halves | 32 lo halves | 16 B pad (geo.hip GC_ROW)
#define RP_WBYTES (256 * RP_ROW)   // 36 864
#define RP_MAXM 256
#define RP_K 32

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
// v_permlane32_swap: a' = [a.lanes0-31, b.lanes0-31], b' = [a.lanes32-63, b.lanes32-63]
__device__ __forceinline__ void swap32(float& a, float& b) {
  const u2 r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// v_permlane16_swap: the odd 16-lane rows of a trade places with the even rows of b
__device__ __forceinline__ void swap16(float& a, float& b) {
  const u2 r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// Persistent workgroups (one per CU) of up to 16 INDEPENDENT waves: a wave owns whole queries (all ceil(n/16) key tiles,
// the softmax included), so there is no workgroup barrier in the loop and the waves of a CU drift apart -- one wave's loads
// and VALU phases sit under another's MFMAs.  Only the A_c image (36 KiB) is shared; each wave has a private LDS slice with
// the folded query as MFMA fragments, qd and its score rows.
// Per key tile, lane = (kx, kg) with kx = lane & 15 and kg = lane >> 4:
//  * basis: lane (kx, kg) runs ONE fp32 Chebyshev recurrence -- scalar kg of key kx (kg = 0: d_idx, 1..3: a_idx[k]) -- and a
//    two-stage permlane swap (a 4 x 4 transpose over the 16-lane rows) hands every lane the orders [8 kg, 8 kg + 8) of all
//    four scalars of its key: exactly a fragment of v_mfma_f32_16x16x32_f16 (K = 32 = the whole expansion).
//  * stage 1, per 16-channel block: 9 MFMAs (3 angular rows x 3 split products) with the COEFFICIENTS as the A operand and the
//    basis as B, so the accumulator comes out transposed -- lane (kx, kg) holds channels 4 kg + r of key kx -- followed by one
//    v_max3 per register (max over the angular rows).
//  * stage 2, per pair of blocks: the 8 maxima of a lane are, as they lie, the A fragment (row = key kx, k slot 8 kg + j) of a
//    second MFMA against the folded query (B: k slot x head) with the channels of a 32-block permuted to match
//    (slot 8 kg + j <-> channel 16 (j >> 2) + 4 kg + (j & 3)); the maxima are split into fp16 hi / lo (v_cvt_pk_f16_f32 +
//    v_fma_mix_f32: 2 instructions per value) and contracted with 3 MFMAs.  The 4 head dots and the reduction over the 256
//    channels -- 256 FMAs, 64 v_max3 and a 60-instruction transposing DPP reduction per tile in the first version of this kernel
//    -- are 24 MFMAs and 128 conversion instructions; the result (keys 4 kg + r, head kx < 4) needs no cross-lane step at all.
//    The query side is scaled per query by a power of two (max |qp| -> [2^13, 2^14)), the coefficient image by 1024; the host
//    checks sum_p |1024 c[ch][p]| < 60 000 (|T_p| <= 1), so no maximum can leave the fp16 range (pem.py geo_cheb_packed).
// fp32 recurrence: 7e-7 worst case on the projected embedding, below the reference's own fp32 sin/cos argument rounding (3e-6).
#define RP_QF_BYTES 4096                       // folded query as stage-2 B fragments: [G 8][kg 4][head 4][8 halves], hi plane | lo plane
#define RP_QW_FLOATS (RP_QF_BYTES / 4 + 128)   // + qd [head][32]

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#ifndef RP_PRIO
#define RP_PRIO 1
#endif

__device__ __forceinline__ unsigned rp_cvt_pk(float a, float b) {
  unsigned r;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// f - (float)half, exact: v_fma_mix_f32 reads the low (HI = 0) or high half of h2 as its first operand
template <int HI>
__device__ __forceinline__ float rp_sub_half(float f, unsigned h2) {
  float r;
  if (HI)
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  else
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h2), "v"(f));
  return r;
}
// (a, b) -> packed fp16 hi pair and lo pair, hi + lo = value to 22 bits (same roundings as sam6d_split_f16)
__device__ __forceinline__ void rp_split2(float a, float b, unsigned& hi, unsigned& lo) {
  hi = rp_cvt_pk(a, b);
  lo = rp_cvt_pk(rp_sub_half<0>(a, hi), rp_sub_half<1>(b, hi));
}
__device__ __forceinline__ half8 rp_h8(unsigned a, unsigned b, unsigned c, unsigned d) {
  return __builtin_bit_cast(half8, u32x4{a, b, c, d});
}

#ifdef RP_STAMP  // diagnostic build only (scratch/): per-wave s_memtime stamps; no output value depends on them
__device__ unsigned long long rp_stamps[4096 * 12];
__device__ unsigned long long rp_phase[4096 * 8];
extern "C" int sam6d_rpe_debug_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rp_stamps), sizeof(unsigned long long) * 4096 * 12);
}
extern "C" int sam6d_rpe_debug_phases(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(rp_phase), sizeof(unsigned long long) * 4096 * 8);
}
#define RP_PH(i) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); ph[i] += now_ - tl; tl = now_; }
#define RP_ST(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define RP_ST(i)
#define RP_PH(i)
#endif

// NP = split products of stage 1.  3: c_hi.t_lo + c_lo.t_hi + c_hi.t_hi, each over all 32 orders.  2: c_hi.t_hi over all 32 orders and
// ONE MFMA for both cross terms of the orders 0..15 (A = [c_lo 0..15 | c_hi 0..15], B = [t_hi 0..15 | t_lo 0..15]: the same image
// read with another per-lane address, the basis fragment one v_permlane32_swap per register) -- the cross terms of the orders >= 16
// are dropped, which the host allows when 2^-10 sum_{p >= 16} |c[ch][p]| is below 3e-8 of the channel's bound (pem.py
// geo_cheb_a_packed: the angular indices live on [0, 12.125], where the coefficients of order 16 are ~1e-6 of the leading ones).
// RAW: P receives the geometric score term itself (4 x n floats per query, listed keys 0, not yet scaled by 1/8) instead of the softmax
// probabilities; q.k^T, the softmax and P.v then happen in sattn_kernel (xattn.hip), and Se is not read.
template <int NP, bool RAW = false>
__global__ __launch_bounds__(768) void rpe_score_kernel(const float4* __restrict__ idx4, const int* __restrict__ pos,
                                                         const float* __restrict__ rows, const unsigned char* __restrict__ Wc,
                                                         const float* __restrict__ qp, const float* __restrict__ qd,
                                                         const float* __restrict__ Se, float* __restrict__ P, int n, int ldp,
                                                         long Q, float xmax_d, float xmax_a, float scale, int mpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* Aw = lds_raw;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int kx = lane & 15, kg = lane >> 4;
  float* qw = reinterpret_cast<float*>(lds_raw + RP_WBYTES) + (size_t)wave * (RP_QW_FLOATS + 4 * mpad);
  unsigned char* qf = reinterpret_cast<unsigned char*>(qw);
  float* qdw = qw + RP_QF_BYTES / 4;
  float* scw = qdw + 128;  // [4][mpad]
#ifdef RP_STAMP
  unsigned long long st[12];
  for (int i = 0; i < 12; ++i) st[i] = 0;
  int nq = 0;
  unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tl = 0;
  st[8] = __builtin_amdgcn_s_memrealtime();
  st[10] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |  // HW_ID | XCC_ID << 32
           ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);
#endif
  RP_ST(0);
  __shared__ int next_query;  // the workgroup's queries (q = blockIdx.x + j * gridDim.x) are handed out in order, one per request
  if (t == 0) next_query = 0;
  for (int i = t; i < RP_WBYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(Aw)[i] = reinterpret_cast<const uint4*>(Wc)[i];
  __syncthreads();
  RP_ST(1);
  const float xmax = kg == 0 ? xmax_d : xmax_a;  // lane row 0 carries the distance index, rows 1..3 the angular ones
  const float uscale = 2.0f / xmax;
  const int ntiles = (n + 15) >> 4;
  const unsigned char* wbase = Aw + (size_t)kx * RP_ROW + kg * 16;
  // second fragment of a block: the lo plane (NP = 3) or the cross-term fragment [lo 0..15 | hi 0..15] (NP = 2)
  const unsigned char* wsec = NP == 3 ? wbase + 64 : Aw + (size_t)kx * RP_ROW + (kg < 2 ? 64 + kg * 16 : (kg - 2) * 16);
  const unsigned char* qfl = qf + (kg * 4 + (kx & 3)) * 16;  // this lane's stage-2 B fragment of 32-block G: + 256 G (+ 2048: lo)
  // Queries are dealt to workgroups round-robin (q % grid) and, inside the workgroup, taken from an LDS counter by whichever wave
  // is free: the SIMD arbitrates its three waves by age, the oldest runs a query in 90 k cycles while the youngest needs 300 k
  // beside it (s_memtime stamps, profiles/README.md), so a static deal leaves the old waves idle at the end.
  // (Measured and removed, round 3: cutting the last my_queries % waves queries of a workgroup -- one or two of 49-50 at B = 32 --
  // into key tiles dealt to all waves, with wave c finishing query c after a barrier: 310.2 / 311.0 us against 309.7 / 310.7 us.)
  const int my_queries = (int)((Q - blockIdx.x + gridDim.x - 1) / gridDim.x);
  auto take = [&]() {
    int j = 0;
    if (lane == 0) j = atomicAdd(&next_query, 1);
    return __builtin_amdgcn_readfirstlane(j);
  };
  // the next query's folded query rows (stage-2 entries e = lane, lane + 64: channels 32 G + 4 kg + {0..3, 16..19} of head e & 3)
  // and qd are requested before the softmax of the current one
  float4 pa[2], pb[2], pqd;
  auto request = [&](long q) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u;
      const float* s = qp + q * 1024 + (e & 3) * 256 + (e >> 4) * 32 + ((e >> 2) & 3) * 4;
      pa[u] = *reinterpret_cast<const float4*>(s);
      pb[u] = *reinterpret_cast<const float4*>(s + 16);
    }
    pqd = *reinterpret_cast<const float4*>(qd + q * 128 + (lane & 31) * 4);
  };
  int jq = take();
  if (jq < my_queries) request(blockIdx.x + (long)jq * gridDim.x);
  while (jq < my_queries) {
    const long q = blockIdx.x + (long)jq * gridDim.x;
#ifdef RP_STAMP
    tl = __builtin_amdgcn_s_memtime();
#endif
    float unscale_a;
    {  // stage the folded query as stage-2 fragments and qd
      float4 a[2] = {pa[0], pa[1]}, b[2] = {pb[0], pb[1]};
      if (lane < 32) reinterpret_cast<float4*>(qdw)[lane] = pqd;
      float am = 0.f;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        am = fmaxf(am, fmaxf(fmaxf(fabsf(a[u].x), fabsf(a[u].y)), fmaxf(fabsf(a[u].z), fabsf(a[u].w))));
        am = fmaxf(am, fmaxf(fmaxf(fabsf(b[u].x), fabsf(b[u].y)), fmaxf(fabsf(b[u].z), fabsf(b[u].w))));
      }
      am = wave_max_dpp(am);
      int k2 = 14 - __builtin_amdgcn_frexp_expf(am);  // am * 2^k2 in [2^13, 2^14)
      k2 = min(max(k2, -100), 100);
      k2 = __builtin_amdgcn_readfirstlane(k2);
      const float beta = ldexpf(1.0f, k2);
      unscale_a = ldexpf(1.0f, -10 - k2);  // the coefficient image carries 1024
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u;
        unsigned h0, h1, h2, h3, l0, l1, l2, l3;
        rp_split2(a[u].x * beta, a[u].y * beta, h0, l0);
        rp_split2(a[u].z * beta, a[u].w * beta, h1, l1);
        rp_split2(b[u].x * beta, b[u].y * beta, h2, l2);
        rp_split2(b[u].z * beta, b[u].w * beta, h3, l3);
        *reinterpret_cast<u32x4*>(qf + e * 16) = u32x4{h0, h1, h2, h3};
        *reinterpret_cast<u32x4*>(qf + 2048 + e * 16) = u32x4{l0, l1, l2, l3};
      }
    }
    // the q.k scores of the query (4 heads x n): in flight under the whole tile loop
    float sev[4][4];
#pragma unroll
    for (int h = 0; h < 4; ++h)
#pragma unroll
      for (int u = 0; u < 4; ++u) sev[h][u] = (!RAW && lane + 64 * u < n) ? Se[(q * 4 + h) * ldp + lane + 64 * u] : 0.f;
    const long pbase = q * n;
    float4 v = idx4[pbase + min(kx, n - 1)];
    int ps = pos[pbase + min(kx, n - 1)];
    RP_PH(0);
    for (int tile = 0; tile < ntiles; ++tile) {
      const int key0 = tile * 16, key = key0 + kx;
      const bool valid = key < n;
      const bool listed = valid && ps >= 0;
      const unsigned listed_mask = (unsigned)(__ballot(listed && kg == 0) & 0xffffull);
      // ---- basis: one recurrence per lane, then the 4 x 4 transpose over the lane rows
      float R[4][8];
      {
        const float x = kg == 0 ? v.x : kg == 1 ? v.y : kg == 2 ? v.z : v.w;
        const bool inside = x >= 0.f && x <= xmax;
        const float u = inside ? fmaf(x, uscale, -1.0f) : 0.0f, u2 = u + u;
        if (tile + 1 < ntiles) {  // next tile's indices: in flight under this tile's arithmetic
          const long pn = pbase + min(key + 16, n - 1);
          v = idx4[pn];
          ps = pos[pn];
        }
        float t0 = 1.0f, t1 = u;
        R[0][0] = t0;
        R[0][1] = t1;
#pragma unroll
        for (int p = 2; p < RP_K; ++p) {
          const float tp = fmaf(u2, t1, -t0);
          t0 = t1;
          t1 = tp;
          R[p >> 3][p & 7] = tp;
        }
        RP_PH(5);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          swap32(R[0][j], R[2][j]);
          swap32(R[1][j], R[3][j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          swap16(R[0][j], R[1][j]);
          swap16(R[2][j], R[3][j]);
        }
      }  // R[G][j] = T_{8 kg + j}(scalar G of key kx)
      RP_PH(6);

      // ---- d part: partial dot over this lane's 8 orders for the 4 heads, summed over the 4 lane rows
      // (Measured and removed, round 3: the same sum as three MFMAs -- A = the distance scalar's basis as it lies, B = qd staged as
      // fp16 hi / lo fragments, the result in the layout of the stage-2 scores: -3 % in the softmax variant of the kernel (308.0
      // against 318.5 us), nothing in the raw variant the pipeline runs (0.2975 / 0.2983 against 0.2963 / 0.2975 ms per launch).)
      float dtot;
      {
        float sd[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const float4 a = *reinterpret_cast<const float4*>(qdw + h * 32 + kg * 8);
          const float4 b = *reinterpret_cast<const float4*>(qdw + h * 32 + kg * 8 + 4);
          float acc = a.x * R[0][0];
          acc = fmaf(a.y, R[0][1], acc);
          acc = fmaf(a.z, R[0][2], acc);
          acc = fmaf(a.w, R[0][3], acc);
          acc = fmaf(b.x, R[0][4], acc);
          acc = fmaf(b.y, R[0][5], acc);
          acc = fmaf(b.z, R[0][6], acc);
          acc = fmaf(b.w, R[0][7], acc);
          sd[h] = acc;
        }
        swap32(sd[0], sd[1]);
        swap32(sd[2], sd[3]);
        float w0 = sd[0] + sd[1], w1 = sd[2] + sd[3];  // lanes 0-31: heads 0 / 2, lanes 32-63: heads 1 / 3
        swap16(w0, w1);
        dtot = w0 + w1;  // lane row kg = 2 b5 + b4 holds head 2 b4 + b5 of key kx
      }

      // ---- basis fragments of the three angular rows (k = 8 kg + j, column = key kx), fp16 hi / lo
      half8 ah[3], al[3];  // NP = 2: al = the cross-term fragment (lanes 0-31: hi of the orders 0..15, lanes 32-63: their lo)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        unsigned h[4], l[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) rp_split2(R[k + 1][2 * j], R[k + 1][2 * j + 1], h[j], l[j]);
        ah[k] = rp_h8(h[0], h[1], h[2], h[3]);
        if (NP == 2) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = __uint_as_float(h[j]), y = __uint_as_float(l[j]);
            swap32(x, y);  // x = [h.lanes 0-31, l.lanes 0-31]
            l[j] = __float_as_uint(x);
          }
        }
        al[k] = rp_h8(l[0], l[1], l[2], l[3]);
      }

      RP_PH(1);
      // ---- contraction.  Software pipeline over the channel blocks, written out in issue order and pinned with sched_barrier:
      // slot s issues the 9 stage-1 MFMAs of block s; between them go the max3 of block s - 1, the fp16 split of block s - 2
      // and (even s >= 4) the 3 stage-2 MFMAs of the block pair (s - 4, s - 3).
      f32x4 acc[2][3];
      half8 bh[2], bl[2];
      float g[2][4];            // maxima of block cb in g[cb & 1]
      unsigned gh[2][4], gl[2][4];  // split maxima of the pair G in [G & 1]: even block in [0..1], odd block in [2..3]
      f32x4 S = {0.f, 0.f, 0.f, 0.f};
      auto load_w = [&](int cb) {  // two blocks ahead of their MFMAs: the LDS latency sits under a whole block of MFMAs
        bh[cb & 1] = *reinterpret_cast<const half8*>(wbase + cb * 16 * RP_ROW);
        bl[cb & 1] = *reinterpret_cast<const half8*>(wsec + cb * 16 * RP_ROW);
      };
      constexpr int NM = 3 * NP;  // stage-1 MFMAs per block
      auto mfma1 = [&](int cb, int i) {  // i-th of the NM MFMAs of block cb: NP = 3: hi.lo, lo.hi, hi.hi per angular row; 2: cross, hi.hi
        const int b = cb & 1, k = i % 3, part = i / 3 + (3 - NP);
        if (part == 0) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[b], al[k], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (part == 1 && NP == 3) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[b], ah[k], acc[b][k], 0, 0, 0);
        if (part == 1 && NP == 2) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[b], al[k], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (part == 2) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[b], ah[k], acc[b][k], 0, 0, 0);
      };
      auto max_row = [&](int cb, int r) {
        const int b = cb & 1;
        g[b][r] = __builtin_fmaxf(__builtin_fmaxf(acc[b][0][r], acc[b][1][r]), acc[b][2][r]);
      };
      auto split_half = [&](int cb, int half) {  // two of the four maxima of block cb
        const int G = cb >> 1, o = 2 * (cb & 1) + half;
        rp_split2(g[cb & 1][2 * half], g[cb & 1][2 * half + 1], gh[G & 1][o], gl[G & 1][o]);
      };
      half8 qh, ql;
      auto load_q = [&](int G) {
        qh = *reinterpret_cast<const half8*>(qfl + G * 256);
        ql = *reinterpret_cast<const half8*>(qfl + G * 256 + 2048);
      };
      auto mfma2 = [&](int G, int i) {
        const int p = G & 1;
        const half8 xh = rp_h8(gh[p][0], gh[p][1], gh[p][2], gh[p][3]);
        if (i == 0) S = __builtin_amdgcn_mfma_f32_16x16x32_f16(rp_h8(gl[p][0], gl[p][1], gl[p][2], gl[p][3]), qh, S, 0, 0, 0);
        if (i == 1) S = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, ql, S, 0, 0, 0);
        if (i == 2) S = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh, qh, S, 0, 0, 0);
      };
      load_w(0);
      load_w(1);
      __builtin_amdgcn_s_setprio(RP_PRIO);  // a wave inside its MFMA loop outranks the waves in their VALU / memory phases
#pragma unroll
      for (int i = 0; i < NM; ++i) mfma1(0, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 1; s <= 18; ++s) {
        if (s + 1 < 16) load_w(s + 1);  // overwrites the fragments of block s - 1, whose MFMAs are all issued
        const bool st2 = (s & 1) == 0 && s >= 4;
        if (st2) load_q((s - 4) >> 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if (s < 16) mfma1(s, i);
          if (NP == 3) {
            if ((i & 1) && s - 1 < 16) max_row(s - 1, i >> 1);          // after MFMAs 1, 3, 5, 7
            if ((i == 2 || i == 6) && s >= 2 && s - 2 < 16) split_half(s - 2, i >> 2);
          } else {
            if ((i == 0 || i == 3) && s >= 2 && s - 2 < 16) split_half(s - 2, i / 3);
            if (i != 0 && i != 3 && s - 1 < 16) max_row(s - 1, i - 1 - (i > 3));  // after MFMAs 1, 2, 4, 5
          }
          if (st2 && i >= NM - 3) mfma2((s - 4) >> 1, i - (NM - 3));
          __builtin_amdgcn_sched_barrier(0);
        }
      }

      __builtin_amdgcn_s_setprio(0);
      RP_PH(2);
      // ---- scores of the tile: the d part from every lane (key kx, head by lane row), then the a part from the lanes kx < 4
      // (head kx, keys 4 kg + r).  Listed keys (outside the Chebyshev range) get 0: their geometric term was added to the q.k
      // scores by rpe_listed_kernel.
      {
        const int h = 2 * (kg & 1) + (kg >> 1);
        scw[h * mpad + key] = listed ? 0.f : dtot;
        if (kx < 4) {
          float4* p = reinterpret_cast<float4*>(scw + kx * mpad + key0 + 4 * kg);
          float4 c = *p;
          const unsigned m = listed_mask >> (4 * kg);
          c.x = (m & 1u) ? 0.f : fmaf(S[0], unscale_a, c.x);
          c.y = (m & 2u) ? 0.f : fmaf(S[1], unscale_a, c.y);
          c.z = (m & 4u) ? 0.f : fmaf(S[2], unscale_a, c.z);
          c.w = (m & 8u) ? 0.f : fmaf(S[3], unscale_a, c.w);
          *p = c;
        }
      }
      RP_PH(3);
    }
    // ---- softmax over the keys (F.softmax: exp(x - max) / sum) of (q.k + geometric term) / 8, probabilities to P[q][h][:]
    jq = take();
    if (jq < my_queries) request(blockIdx.x + (long)jq * gridDim.x);
    if constexpr (RAW) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        float* pr = P + (q * 4 + h) * ldp;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = lane + 64 * u;
          if (j < n) pr[j] = scw[h * mpad + j];
        }
      }
    } else
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      float xv[4];
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        xv[u] = j < n ? (scw[h * mpad + j] + sev[h][u]) * scale : -INFINITY;
        mx = fmaxf(mx, xv[u]);
      }
      mx = wave_max_dpp(mx);
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        // (v_exp_f32 on the log2-scaled difference: 1 ulp, arguments <= 0; expf's range reduction is 10 instructions per value)
        xv[u] = lane + 64 * u < n ? __builtin_amdgcn_exp2f((xv[u] - mx) * 1.4426950408889634f) : 0.f;
        sum += xv[u];
      }
      sum = wave_sum_dpp(sum);
      const float inv = 1.0f / sum;
      float* pr = P + (q * 4 + h) * ldp;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < n) pr[j] = xv[u] * inv;
      }
    }
    RP_PH(4);
#ifdef RP_STAMP
    if (nq < 6) st[2 + nq] = __builtin_amdgcn_s_memtime();
    ++nq;
#endif
  }
#ifdef RP_STAMP
  st[9] = __builtin_amdgcn_s_memrealtime();
  st[11] = nq;
  if (lane == 0)
  {
    for (int i = 0; i < 12; ++i) rp_stamps[((size_t)blockIdx.x * 12 + wave) * 12 + i] = st[i];
    for (int i = 0; i < 8; ++i) rp_phase[((size_t)blockIdx.x * 12 + wave) * 8 + i] = ph[i];
  }
#endif
}

// Geometric score term of the listed pairs (an index outside [0, xmax]: the bg token, 2n-1 of n^2 pairs): one wave per pair
// dots its stored bias-free embedding row with the 4 folded queries and adds the result to the q.k score of that (query, head,
// key).  Runs between the q.k^T GEMM and rpe_score_kernel; each score element is touched by at most one wave.
__global__ __launch_bounds__(256) void rpe_listed_kernel(const int* __restrict__ list, const float* __restrict__ rows,
                                                         const float* __restrict__ qp, float* __restrict__ Se, int n, int ldp) {
  const int lane = threadIdx.x & 63;
  const int count = list[0];
  const int stride = gridDim.x * 4;
  for (int i = blockIdx.x * 4 + (threadIdx.x >> 6); i < count; i += stride) {
    const int e = list[1 + i];
    const long q = e / n;
    const int m = e - (int)q * n;
    const float4 r = *reinterpret_cast<const float4*>(rows + (size_t)i * 256 + lane * 4);
    const float* qpq = qp + q * 1024 + lane * 4;
    float sp[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float4 w = *reinterpret_cast<const float4*>(qpq + h * 256);
      sp[h] = wave_sum_dpp((w.x * r.x + w.y * r.y) + (w.z * r.z + w.w * r.w));
    }
    if (lane < 4) {
      const float mine = lane == 0 ? sp[0] : lane == 1 ? sp[1] : lane == 2 ? sp[2] : sp[3];
      Se[(q * 4 + lane) * ldp + m] += mine;
    }
  }
}

extern "C" int sam6d_rpe_scores(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb,
                                float xmax, const float* qp, const float* qd, float* qk, float* P, long Q, int n, int ldp,
                                void* stream) {
  return sam6d_rpe_scores2(idx_ws, pos_ws, list_ws, rows, wa_cheb, xmax, xmax, 3, qp, qd, qk, P, Q, n, ldp, stream);
}

static int rpe_scores_impl(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb,
                           float xmax, float xmax_a, int products, const float* qp, const float* qd, float* qk, float* P, long Q, int n,
                           int ldp, bool raw, void* stream);
extern "C" int sam6d_rpe_scores2(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb,
                                 float xmax, float xmax_a, int products, const float* qp, const float* qd, float* qk, float* P, long Q,
                                 int n, int ldp, void* stream) {
  SAM6D_REQUIRE(qk, "rpe_scores: null pointer");
  return rpe_scores_impl(idx_ws, pos_ws, list_ws, rows, wa_cheb, xmax, xmax_a, products, qp, qd, qk, P, Q, n, ldp, false, stream);
}
extern "C" int sam6d_rpe_geo_scores(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb,
                                    float xmax, float xmax_a, int products, const float* qp, const float* qd, float* G, long Q, int n,
                                    int ldp, void* stream) {
  return rpe_scores_impl(idx_ws, pos_ws, list_ws, rows, wa_cheb, xmax, xmax_a, products, qp, qd, nullptr, G, Q, n, ldp, true, stream);
}

// raw = false: qk holds q.k^T, the listed pairs' terms are added to it first, P = softmax probabilities.
// raw = true:  P = the geometric score term alone (the listed pairs' terms are added to it AFTER the score kernel has written its zeros)
static int rpe_scores_impl(const float* idx_ws, const int* pos_ws, const int* list_ws, const float* rows, const void* wa_cheb,
                           float xmax, float xmax_a, int products, const float* qp, const float* qd, float* qk, float* P, long Q, int n,
                           int ldp, bool raw, void* stream) {
  SAM6D_REQUIRE(idx_ws && pos_ws && list_ws && rows && wa_cheb && qp && qd && P, "rpe_scores: null pointer");
  SAM6D_REQUIRE(products == 2 || products == 3, "rpe_scores: products must be 2 or 3 (got %d)", products);
  SAM6D_REQUIRE(xmax_a > 0.f, "rpe_scores: xmax_a must be positive");
  SAM6D_REQUIRE(Q >= 0 && n > 0 && n <= RP_MAXM && ldp >= n, "rpe_scores: need 0 < n <= %d and ldp >= n (n = %d, ldp = %d)",
                RP_MAXM, n, ldp);
  SAM6D_REQUIRE(xmax > 0.f, "rpe_scores: xmax must be positive");
  SAM6D_REQUIRE((((size_t)idx_ws | (size_t)wa_cheb | (size_t)qp | (size_t)qd | (size_t)rows) & 15) == 0,
                "rpe_scores: idx_ws / wa_cheb / qp / qd / rows must be 16-byte aligned");
  if (Q == 0) return 0;
  const int lds_max = 160 * 1024 - 64;  // dynamic part: the kernel has one static word (its query counter)
  static int n_cu_dev[SAM6D_MAX_DEVICES];
  static unsigned long long rpe_done = 0;
  int dev = 0;
  if (sam6d_first_use_on_device(&rpe_done, &dev)) {
    SAM6D_REQUIRE(dev >= 0, "rpe_scores: device ordinal beyond SAM6D_MAX_DEVICES");
    int cu = 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rpe_score_kernel<3>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(rpe_score_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(rpe_score_kernel<3, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(rpe_score_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      sam6d_set_error("rpe_scores: cannot reserve %d bytes of LDS / query the device: %s", lds_max, hipGetErrorString(e));
      return e != hipSuccess ? (int)e : SAM6D_EINVAL;
    }
    n_cu_dev[dev] = cu;
    sam6d_setup_done_on_device(&rpe_done);
  }
  const int n_cu = n_cu_dev[dev];
  const float scale = 0.125f;  // 1/sqrt(64): d_model 256, 4 heads (coarse_point_matching.py:24, fine_point_matching.py:31)
  const int mpad = ((n + 15) / 16) * 16;
  const int per_wave = (RP_QW_FLOATS + 4 * mpad) * 4;
  int waves = (lds_max - RP_WBYTES) / per_wave;  // 16 for n <= 208, 15 up to 256
  if (waves > 12) waves = 12;  // 3 waves per SIMD: 170 registers per lane, no spills
  long blocks = (Q + waves - 1) / waves;
  if (blocks > n_cu) blocks = n_cu;
  long lblocks = (Q * 2 + 3) / 4;  // about two listed pairs per query token (its bg key and its share of the bg query)
  if (lblocks > 4096) lblocks = 4096;
  const dim3 sgrid((unsigned)blocks), sblock(64 * waves);
  const size_t slds = RP_WBYTES + waves * per_wave;
  const float4* idx4 = reinterpret_cast<const float4*>(idx_ws);
  const unsigned char* wc = reinterpret_cast<const unsigned char*>(wa_cheb);
  if (raw) {
    if (products == 3)
      hipLaunchKernelGGL((rpe_score_kernel<3, true>), sgrid, sblock, slds, (hipStream_t)stream, idx4, pos_ws, rows, wc, qp, qd,
                         (const float*)nullptr, P, n, ldp, Q, xmax, xmax_a, scale, mpad);
    else
      hipLaunchKernelGGL((rpe_score_kernel<2, true>), sgrid, sblock, slds, (hipStream_t)stream, idx4, pos_ws, rows, wc, qp, qd,
                         (const float*)nullptr, P, n, ldp, Q, xmax, xmax_a, scale, mpad);
    SAM6D_LAUNCH_CHECK_CONT("rpe_geo_scores");
    hipLaunchKernelGGL(rpe_listed_kernel, dim3((unsigned)lblocks), dim3(256), 0, (hipStream_t)stream, list_ws, rows, qp, P, n, ldp);
    SAM6D_LAUNCH_CHECK("rpe_geo_scores(listed pairs)");
  }
  hipLaunchKernelGGL(rpe_listed_kernel, dim3((unsigned)lblocks), dim3(256), 0, (hipStream_t)stream, list_ws, rows, qp, qk, n, ldp);
  SAM6D_LAUNCH_CHECK_CONT("rpe_scores(listed pairs)");
  if (products == 3)
    hipLaunchKernelGGL(rpe_score_kernel<3>, dim3((unsigned)blocks), dim3(64 * waves), RP_WBYTES + waves * per_wave,
                       (hipStream_t)stream, reinterpret_cast<const float4*>(idx_ws), pos_ws, rows,
                       reinterpret_cast<const unsigned char*>(wa_cheb), qp, qd, qk, P, n, ldp, Q, xmax, xmax_a, scale, mpad);
  else
    hipLaunchKernelGGL(rpe_score_kernel<2>, dim3((unsigned)blocks), dim3(64 * waves), RP_WBYTES + waves * per_wave,
                       (hipStream_t)stream, reinterpret_cast<const float4*>(idx_ws), pos_ws, rows,
                       reinterpret_cast<const unsigned char*>(wa_cheb), qp, qd, qk, P, n, ldp, Q, xmax, xmax_a, scale, mpad);
  SAM6D_LAUNCH_CHECK("rpe_scores");
}

// dst[b][c][j] = src[b][j][c]  (values v of the attention as the N x K operand of the P.V GEMM)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, long lds_, long ss, int n, int ncol,
                                                        float* __restrict__ dst, long ldd, long sd) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z, j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 8 rows per pass
  const float* s = src + (size_t)b * ss;
  float* d = dst + (size_t)b * sd;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = j0 + ty + 8 * u, c = c0 + tx;
    tile[ty + 8 * u][tx] = (j < n && c < ncol) ? s[(size_t)j * lds_ + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int c = c0 + ty + 8 * u, j = j0 + tx;
    if (c < ncol && j < n) d[(size_t)c * ldd + j] = tile[tx][ty + 8 * u];
  }
}

extern "C" int sam6d_transpose(const float* src, long ld_src, long stride_src, int B, int n, int ncol, float* dst, long ld_dst,
                               long stride_dst, void* stream) {
  SAM6D_REQUIRE(src && dst, "transpose: null pointer");
  SAM6D_REQUIRE(B >= 0 && B <= 65535 && n > 0 && ncol > 0 && ld_src >= ncol && ld_dst >= n, "transpose: bad sizes");
  if (B == 0) return 0;
  hipLaunchKernelGGL(transpose_kernel, dim3((n + 31) / 32, (ncol + 31) / 32, B), dim3(256), 0, (hipStream_t)stream, src, ld_src,
                     stride_src, n, ncol, dst, ld_dst, stride_dst);
  SAM6D_LAUNCH_CHECK("transpose");
}
