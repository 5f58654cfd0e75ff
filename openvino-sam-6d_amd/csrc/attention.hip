// Softmax attention of the sparse (197-token) transformer layers for gfx950.
//
//   RPE self layer  (PEM/model/transformer.py:366-420):  s[h,n,m] = (q_h[n].k_h[m] + q_h[n].proj_p(E[n,m])_h) / sqrt(64)
//   cross layer     (PEM/model/transformer.py:95-150)  :  s[h,n,m] =  q_h[n].k_h[m] / sqrt(64)
//   out[n, h*64+c] = sum_m softmax_m(s)[h,n,m] v[m, h*64+c]
//
// proj_p is folded into the query: q_h.(W_p E + b_p)_h = (W_p,h^T q_h).E + q_h.b_p,h.  The second term
// does not depend on m and cancels in the softmax, so it is dropped; qp[n,h,:] = W_p,h^T q_h (4 x 256 per token) is
// produced by the GEMM kernel.  Executed flops per layer fall from 5.09 GFLOP to 0.08 GFLOP per proposal and the
// RPE kernel becomes an HBM stream over E (39.7 MB per proposal, read once per layer).
//
// One workgroup = AT_Q consecutive query tokens of one cloud, 4 waves.  Every key/value row fetched from L2 is used for
// AT_Q queries (the one-query form moved 400 KB of k/v per query through L2: 22 TB/s aggregate, the L2 limit).
//   Phase 1: waves stride over the key rows; a lane owns 4 channels (16-byte loads, a 1 KiB row per wave-instruction).
//            q.k : AT_Q partial dots per lane, reduced inside the 16-lane head group by a transpose butterfly.
//            qp.E: the query's own embedding row (streamed, non-temporal), 4 per-head partial dots per lane, folded
//                  with a transpose butterfly across the wave (xor 32, xor 16) and then inside the head group.
//   Phase 2: softmax over the keys in LDS, one (query, head) row per wave at a time (F.softmax: exp(x-max)/sum).
//   Phase 3: thread (h,c) accumulates sum_m p[q][h][m] v[m,h,c] for the AT_Q queries from one coalesced read of v.
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef float fx4 __attribute__((ext_vector_type(4)));
#define AT_MAXM 256
#define AT_Q 4

__device__ __forceinline__ float dot4(const float4 a, const float4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

template <bool RPE>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ qp,
                                                        const float* __restrict__ E, float* __restrict__ out, int n,
                                                        int m, long ldq, long ldk, long ldv, long ldo, long sq, long sk,
                                                        long sv, long so, float scale) {
  __shared__ float s_s[AT_Q][4][AT_MAXM];
  const int b = blockIdx.y, i0 = blockIdx.x * AT_Q;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nq = min(AT_Q, n - i0);  // queries handled by this workgroup (the last one may be partial)
  float4 ql[AT_Q];                   // lane's 4 channels of each query (head = lane >> 4)
  float4 qf[AT_Q][4];                // RPE: folded queries, [query][head]
#pragma unroll
  for (int u = 0; u < AT_Q; ++u) {
    const int i = min(i0 + u, n - 1);
    ql[u] = *reinterpret_cast<const float4*>(q + (size_t)b * sq + (size_t)i * ldq + lane * 4);
    if (RPE) {
      const float* qpr = qp + ((size_t)b * n + i) * 1024;
#pragma unroll
      for (int h = 0; h < 4; ++h) qf[u][h] = *reinterpret_cast<const float4*>(qpr + h * 256 + lane * 4);
    }
  }
  const float* kb = k + (size_t)b * sk;
  const bool hi32 = lane & 32, hi16 = lane & 16, hi8 = lane & 8, hi4 = lane & 4;
  for (int j = wave; j < m; j += 4) {
    const float4 kv = *reinterpret_cast<const float4*>(kb + (size_t)j * ldk + lane * 4);
    float4 ev[AT_Q];
    if (RPE) {
#pragma unroll
      for (int u = 0; u < AT_Q; ++u) {
        const int i = min(i0 + u, n - 1);
        const fx4 e = __builtin_nontemporal_load(
            reinterpret_cast<const fx4*>(E + (((size_t)b * n + i) * (size_t)m + j) * 256 + lane * 4));
        ev[u] = make_float4(e.x, e.y, e.z, e.w);
      }
    }
    // q.k partials of the AT_Q queries (per 16-lane head group) + RPE term folded to the same head group
    float se[AT_Q];
#pragma unroll
    for (int u = 0; u < AT_Q; ++u) {
      se[u] = dot4(ql[u], kv);
      if (RPE) {
        float sp[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) sp[h] = dot4(qf[u][h], ev[u]);
        // transpose butterfly over the wave: heads {0,1} stay in lanes 0-31, {2,3} in 32-63; then even/odd head by bit 4
        const float s0 = hi32 ? sp[0] : sp[2], s1 = hi32 ? sp[1] : sp[3];
        const float k0 = hi32 ? sp[2] : sp[0], k1 = hi32 ? sp[3] : sp[1];
        const float pa = k0 + __shfl_xor(s0, 32, 64), pb = k1 + __shfl_xor(s1, 32, 64);
        const float send = hi16 ? pa : pb, keep = hi16 ? pb : pa;
        se[u] += keep + __shfl_xor(send, 16, 64);  // lane group (lane>>4) carries head (lane>>4) of both terms
      }
    }
    // reduce the 4 query values inside each 16-lane group with a transpose butterfly: 2 + 1 + 2 shuffles instead of 16;
    // afterwards the lanes with (lane & 3) == 0 hold the group sum of query 2*bit3 + bit2
    {
      const float a0 = hi8 ? se[0] : se[2], a1 = hi8 ? se[1] : se[3];  // the pair the partner keeps
      const float b0 = hi8 ? se[2] : se[0], b1 = hi8 ? se[3] : se[1];
      const float c0 = b0 + __shfl_xor(a0, 8, 64), c1 = b1 + __shfl_xor(a1, 8, 64);  // bit3=0: queries {0,1}; bit3=1: {2,3}
      const float snd = hi4 ? c0 : c1, kp = hi4 ? c1 : c0;                            // bit2 selects the odd query
      float r = kp + __shfl_xor(snd, 4, 64);
      r += __shfl_xor(r, 1, 64);
      r += __shfl_xor(r, 2, 64);
      const int qi = ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1);
      if ((lane & 3) == 0 && qi < nq) s_s[qi][lane >> 4][j] = r * scale;
    }
  }
  __syncthreads();
  for (int row = wave; row < nq * 4; row += 4) {  // softmax of one (query, head) row per wave
    float* sr = s_s[row >> 2][row & 3];
    float mx = -INFINITY;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, sr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < m; j += 64) {
      const float e = expf(sr[j] - mx);
      sr[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < m; j += 64) sr[j] *= inv;
  }
  __syncthreads();
  const float* vb = v + (size_t)b * sv + t;
  const int h = t >> 6;
  float acc[AT_Q];
#pragma unroll
  for (int u = 0; u < AT_Q; ++u) acc[u] = 0.f;
  for (int j = 0; j < m; ++j) {
    const float vv = vb[(size_t)j * ldv];
#pragma unroll
    for (int u = 0; u < AT_Q; ++u)
      if (u < nq) acc[u] = fmaf(s_s[u][h][j], vv, acc[u]);
  }
#pragma unroll
  for (int u = 0; u < AT_Q; ++u)
    if (u < nq) out[(size_t)b * so + (size_t)(i0 + u) * ldo + t] = acc[u];
}

extern "C" int sam6d_attention(const float* q, const float* k, const float* v, const float* qp, const float* E, float* out,
                               int B, int n, int m, long ldq, long ldk, long ldv, long ldo, long sq, long sk, long sv,
                               long so, void* stream) {
  SAM6D_REQUIRE(q && k && v && out, "attention: null pointer");
  SAM6D_REQUIRE((qp == nullptr) == (E == nullptr), "attention: qp and E must be given together (RPE) or both NULL");
  SAM6D_REQUIRE(B >= 0 && n > 0 && m > 0 && m <= AT_MAXM, "attention: need 0 < m <= %d (got %d)", AT_MAXM, m);
  SAM6D_REQUIRE(B <= 65535, "attention: B must be <= 65535");
  SAM6D_REQUIRE(((ldq | ldk | ldv | ldo | sq | sk | sv | so) & 3) == 0, "attention: strides must be multiples of 4 floats");
  if (B == 0) return 0;
  const float scale = 0.125f;  // 1/sqrt(64): d_model 256, 4 heads (coarse_point_matching.py:24, fine_point_matching.py:31)
  dim3 grid(cdiv(n, AT_Q), B);
  if (E)
    hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, qp, E, out, n, m, ldq,
                       ldk, ldv, ldo, sq, sk, sv, so, scale);
  else
    hipLaunchKernelGGL(attention_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, qp, E, out, n, m, ldq,
                       ldk, ldv, ldo, sq, sk, sv, so, scale);
  SAM6D_LAUNCH_CHECK("attention");
}
