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
// v_mfma_f32_16x16x32_f16 (fp16 hi/lo split, 3 products), followed by max over k, the 4 head dots on packed-fp32 VALU and a
// transposing DPP reduction over the 16 channel lanes.  Pairs outside [0, xmax] (the bg token: 2n-1 of n^2) take their
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
// the folded query [channel][head], qd and its score rows.
// Per key tile, lane = (kx, kg) with kx = lane & 15 (key of the tile for the A operand, channel of the block for B / D) and
// kg = lane >> 4:
//  * basis: lane (kx, kg) runs ONE fp32 Chebyshev recurrence -- scalar kg of key kx (kg = 0: d_idx, 1..3: a_idx[k]) -- and a
//    two-stage permlane swap (a 4 x 4 transpose over the 16-lane rows) hands every lane the orders [8 kg, 8 kg + 8) of all
//    four scalars of its key: exactly the A fragment of v_mfma_f32_16x16x32_f16 (K = 32 = the whole expansion).
//  * contraction: per 16-channel block 9 MFMAs (3 angular rows x 3 split products), max over the angular rows, 4 head dots
//    as packed FMAs against the folded query from LDS; D layout: lane = channel kx, rows = keys 4 kg + r.
//  * a transposing DPP reduction over the 16 channel lanes leaves one (key, head) total per lane.
// fp32 recurrence: 7e-7 worst case on the projected embedding, below the reference's own fp32 sin/cos argument rounding (3e-6).
#define RP_QW_FLOATS (256 * 4 + 128)  // per-wave: folded query / 1024 as [channel][head], then qd [head][32]

__global__ __launch_bounds__(768) void rpe_score_kernel(const float4* __restrict__ idx4, const int* __restrict__ pos,
                                                         const float* __restrict__ rows, const unsigned char* __restrict__ Wc,
                                                         const float* __restrict__ qp, const float* __restrict__ qd,
                                                         const float* __restrict__ Se, float* __restrict__ P, int n, int ldp,
                                                         long Q, float xmax, float scale, int mpad) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned char* Aw = lds_raw;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, nwaves = blockDim.x >> 6;
  const int kx = lane & 15, kg = lane >> 4;
  float* qw = reinterpret_cast<float*>(lds_raw + RP_WBYTES) + (size_t)wave * (RP_QW_FLOATS + 4 * mpad);
  float* qdw = qw + 1024;
  float* scw = qdw + 128;  // [4][mpad]
  for (int i = t; i < RP_WBYTES / 16; i += blockDim.x) reinterpret_cast<uint4*>(Aw)[i] = reinterpret_cast<const uint4*>(Wc)[i];
  __syncthreads();
  const float uscale = 2.0f / xmax;
  const int ntiles = (n + 15) >> 4;
  const unsigned char* wbase = Aw + (size_t)kx * RP_ROW + kg * 16;
  // query q -> workgroup q % grid, wave (q / grid) % nwaves: the last, partial round is spread over all CUs
  const long qstride = (long)gridDim.x * nwaves;
  for (long q = blockIdx.x + (long)gridDim.x * wave; q < Q; q += qstride) {
    {  // stage the folded query (coefficient image is scaled by 1024, a power of two: undone here, exactly) and qd
      const float* s = qp + q * 1024 + lane * 4;
      const float un = 1.0f / 1024.0f;
      const float4 a0 = *reinterpret_cast<const float4*>(s), a1 = *reinterpret_cast<const float4*>(s + 256);
      const float4 a2 = *reinterpret_cast<const float4*>(s + 512), a3 = *reinterpret_cast<const float4*>(s + 768);
      float4* d = reinterpret_cast<float4*>(qw) + lane * 4;
      d[0] = make_float4(a0.x * un, a1.x * un, a2.x * un, a3.x * un);
      d[1] = make_float4(a0.y * un, a1.y * un, a2.y * un, a3.y * un);
      d[2] = make_float4(a0.z * un, a1.z * un, a2.z * un, a3.z * un);
      d[3] = make_float4(a0.w * un, a1.w * un, a2.w * un, a3.w * un);
      if (lane < 32) reinterpret_cast<float4*>(qdw)[lane] = *reinterpret_cast<const float4*>(qd + q * 128 + lane * 4);
    }
    const long pbase = q * n;
    float4 v = idx4[pbase + min(kx, n - 1)];
    int ps = pos[pbase + min(kx, n - 1)];
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

      // ---- d part: partial dot over this lane's 8 orders for the 4 heads, summed over the 4 lane rows
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
        const float tot = w0 + w1;  // lane row kg = 2 b5 + b4 holds head 2 b4 + b5
        const int h = 2 * (kg & 1) + (kg >> 1);
        if (valid && !listed) scw[h * mpad + key] = tot;
      }

      // ---- A fragments of the three angular rows (row = key kx, k = 8 kg + j), fp16 hi / lo
      half8 ah[3], al[3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float f = R[k + 1][j];
          const _Float16 h = (_Float16)f;
          ah[k][j] = h;
          al[k][j] = (_Float16)(f - (float)h);
        }

      // ---- contraction over the 16 channel blocks; s01 / s23[r] = partial head dots of key row 4 kg + r over this lane's channels
      f2 s01[4], s23[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        s01[r] = f2{0.f, 0.f};
        s23[r] = f2{0.f, 0.f};
      }
      const float* qbase = qw + kx * 4;
      // Software pipeline over the channel blocks, written out in issue order and pinned with sched_barrier: the 9 MFMAs of
      // block cb + 1 alternate with the 12 VALU instructions that consume block cb (a 16x16x32 MFMA holds the vector issue
      // port for 8 of its 16 cycles; in program order behind a blocked MFMA a wave's own VALU work could not use the rest).
      f32x4 acc[2][3];
      half8 bh[2], bl[2];
      float4 qv[3];
      auto load_w = [&](int cb) {  // two blocks ahead of their MFMAs: the LDS latency sits under a whole block of MFMAs
        bh[cb & 1] = *reinterpret_cast<const half8*>(wbase + cb * 16 * RP_ROW);
        bl[cb & 1] = *reinterpret_cast<const half8*>(wbase + cb * 16 * RP_ROW + 64);
        qv[cb % 3] = *reinterpret_cast<const float4*>(qbase + cb * 64);
      };
      auto mfma1 = [&](int cb, int i) {  // i-th of the 9 MFMAs of block cb: products lo.hi, hi.lo, hi.hi per angular row
        const int b = cb & 1, k = i % 3, part = i / 3;
        if (part == 0) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[k], bh[b], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (part == 1) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bl[b], acc[b][k], 0, 0, 0);
        if (part == 2) acc[b][k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bh[b], acc[b][k], 0, 0, 0);
      };
      auto consume = [&](int cb, int r) {  // max over the angular rows, 4 head dots of key row 4 kg + r
        const int b = cb & 1;
        const float g = __builtin_fmaxf(__builtin_fmaxf(acc[b][0][r], acc[b][1][r]), acc[b][2][r]);
        // four plain FMAs: v_pk_fma_f32 issues at half rate on gfx950 (no gain) and packed fp32 is avoided library-wide
        s01[r].x = fmaf(g, qv[cb % 3].x, s01[r].x);
        s01[r].y = fmaf(g, qv[cb % 3].y, s01[r].y);
        s23[r].x = fmaf(g, qv[cb % 3].z, s23[r].x);
        s23[r].y = fmaf(g, qv[cb % 3].w, s23[r].y);
      };
      load_w(0);
      load_w(1);
#pragma unroll
      for (int i = 0; i < 9; ++i) mfma1(0, i);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int cb = 0; cb < 16; ++cb) {
        if (cb + 1 < 16) {
          if (cb + 2 < 16) load_w(cb + 2);  // overwrites the weight registers of block cb, whose MFMAs are all issued
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 9; ++i) {
            mfma1(cb + 1, i);
            if (i & 1) consume(cb, i >> 1);  // after MFMAs 1, 3, 5, 7: rows 0..3
            __builtin_amdgcn_sched_barrier(0);
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) consume(cb, r);
        }
      }

      // ---- transposing reduction over the 16 channel lanes of each lane row: 16 values -> 1 per lane
      // value index = h * 4 + r; each stage pairs (2i, 2i+1) and keeps the one selected by a lane bit, so after the four
      // stages lane bits (b3 b2 b1 b0) hold index b3 + 2 b2 + 4 b1 + 8 b0:  r = b3 + 2 b2,  h = b1 + 2 b0
      {
        float V[16];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          V[r] = s01[r].x;
          V[4 + r] = s01[r].y;
          V[8 + r] = s23[r].x;
          V[12 + r] = s23[r].y;
        }
        const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
        float W2[8], W3[4], W4[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float keep = b3 ? V[2 * i + 1] : V[2 * i], send = b3 ? V[2 * i] : V[2 * i + 1];
          W2[i] = keep + dpp_mov<0x140>(send);  // row_mirror: lane i <-> 15 - i
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float keep = b2 ? W2[2 * i + 1] : W2[2 * i], send = b2 ? W2[2 * i] : W2[2 * i + 1];
          W3[i] = keep + dpp_mov<0x141>(send);  // row_half_mirror: lane i <-> 7 - i
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float keep = b1 ? W3[2 * i + 1] : W3[2 * i], send = b1 ? W3[2 * i] : W3[2 * i + 1];
          W4[i] = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1]
        }
        const float keep = b0 ? W4[1] : W4[0], send = b0 ? W4[0] : W4[1];
        const float z = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]
        const int r = (b3 ? 1 : 0) + (b2 ? 2 : 0), h = (b1 ? 1 : 0) + (b0 ? 2 : 0);
        const int row = 4 * kg + r;
        if (key0 + row < n && !((listed_mask >> row) & 1u)) scw[h * mpad + key0 + row] += z;
      }

      // listed keys (outside the Chebyshev range): their geometric term was added to the q.k scores by rpe_listed_kernel
      if (listed && kg == 0) {
#pragma unroll
        for (int h = 0; h < 4; ++h) scw[h * mpad + key] = 0.f;
      }
    }
    // ---- softmax over the keys (F.softmax: exp(x - max) / sum) of (q.k + geometric term) / 8, probabilities to P[q][h][:]
#pragma unroll 1
    for (int h = 0; h < 4; ++h) {
      const float* se = Se + (q * 4 + h) * ldp;
      float xv[4];
      float mx = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        xv[u] = j < n ? (scw[h * mpad + j] + se[j]) * scale : -INFINITY;
        mx = fmaxf(mx, xv[u]);
      }
      mx = wave_max_dpp(mx);
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        xv[u] = lane + 64 * u < n ? expf(xv[u] - mx) : 0.f;
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
  }
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
  SAM6D_REQUIRE(idx_ws && pos_ws && list_ws && rows && wa_cheb && qp && qd && qk && P, "rpe_scores: null pointer");
  SAM6D_REQUIRE(Q >= 0 && n > 0 && n <= RP_MAXM && ldp >= n, "rpe_scores: need 0 < n <= %d and ldp >= n (n = %d, ldp = %d)",
                RP_MAXM, n, ldp);
  SAM6D_REQUIRE(xmax > 0.f, "rpe_scores: xmax must be positive");
  SAM6D_REQUIRE((((size_t)idx_ws | (size_t)wa_cheb | (size_t)qp | (size_t)qd | (size_t)rows) & 15) == 0,
                "rpe_scores: idx_ws / wa_cheb / qp / qd / rows must be 16-byte aligned");
  if (Q == 0) return 0;
  const int lds_max = 160 * 1024;
  static int n_cu_dev[SAM6D_MAX_DEVICES];
  static unsigned long long rpe_done = 0;
  int dev = 0;
  if (sam6d_first_use_on_device(&rpe_done, &dev)) {
    int cu = 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(rpe_score_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_max);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      sam6d_set_error("rpe_scores: cannot reserve %d bytes of LDS / query the device: %s", lds_max, hipGetErrorString(e));
      return e != hipSuccess ? (int)e : SAM6D_EINVAL;
    }
    n_cu_dev[dev & 63] = cu;
  }
  const int n_cu = n_cu_dev[dev & 63];
  const float scale = 0.125f;  // 1/sqrt(64): d_model 256, 4 heads (coarse_point_matching.py:24, fine_point_matching.py:31)
  const int mpad = ((n + 15) / 16) * 16;
  const int per_wave = (RP_QW_FLOATS + 4 * mpad) * 4;
  int waves = (lds_max - RP_WBYTES) / per_wave;  // 16 for n <= 208, 15 up to 256
  if (waves > 12) waves = 12;  // 3 waves per SIMD: 170 registers per lane, no spills
  long blocks = (Q + waves - 1) / waves;
  if (blocks > n_cu) blocks = n_cu;
  long lblocks = (Q * 2 + 3) / 4;  // about two listed pairs per query token (its bg key and its share of the bg query)
  if (lblocks > 4096) lblocks = 4096;
  hipLaunchKernelGGL(rpe_listed_kernel, dim3((unsigned)lblocks), dim3(256), 0, (hipStream_t)stream, list_ws, rows, qp, qk, n, ldp);
  SAM6D_LAUNCH_CHECK_CONT("rpe_scores(listed pairs)");
  hipLaunchKernelGGL(rpe_score_kernel, dim3((unsigned)blocks), dim3(64 * waves), RP_WBYTES + waves * per_wave,
                     (hipStream_t)stream, reinterpret_cast<const float4*>(idx_ws), pos_ws, rows,
                     reinterpret_cast<const unsigned char*>(wa_cheb), qp, qd, qk, P, n, ldp, Q, xmax, scale, mpad);
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
