// Softmax attention of the sparse (197-token) transformer layers for gfx950.
//
//   RPE self layer  (PEM/model/transformer.py:366-420):  s[h,n,m] = (q_h[n].k_h[m] + q_h[n].proj_p(E[n,m])_h) / sqrt(64)
//   cross layer     (PEM/model/transformer.py:95-150)  :  s[h,n,m] =  q_h[n].k_h[m] / sqrt(64)
//   out[n, h*64+c] = sum_m softmax_m(s)[h,n,m] v[m, h*64+c]
//
// proj_p is folded into the query: q_h.(W_p E + b_p)_h = (W_p,h^T q_h).E + q_h.b_p,h.  The second term
// does not depend on m and cancels in the softmax, so it is dropped; qp[n,h,:] = W_p,h^T q_h (4 x 256 per token) is
// produced by the GEMM kernel.  Executed flops per layer fall from 5.09 GFLOP to 0.08 GFLOP per proposal and the
// kernel becomes a pure HBM stream over E (39.7 MB per proposal, read once per layer).
//
// One workgroup per (b, n) query token, 4 waves.  Phase 1: the 197 embedding rows of that token are streamed with
// 16-byte lanes (one 1 KiB row per wave-instruction, fully coalesced), each row reduced against the 4 folded queries
// by a wave butterfly.  Phase 2: per-head softmax in LDS (one wave per head).  Phase 3: thread (h,c) accumulates
// sum_m p[h,m] v[m,h,c] with coalesced reads of v rows (L2-resident: 201 KB per proposal).
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef float fx4 __attribute__((ext_vector_type(4)));
#define AT_MAXM 256
template <bool RPE>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                        const float* __restrict__ v, const float* __restrict__ qp,
                                                        const float* __restrict__ E, float* __restrict__ out, int n,
                                                        int m, long ldq, long ldk, long ldv, long ldo, long sq, long sk,
                                                        long sv, long so, float scale) {
  __shared__ float s_s[4][AT_MAXM];
  __shared__ float s_q[256];
  const int b = blockIdx.y, i = blockIdx.x;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* qrow = q + (size_t)b * sq + (size_t)i * ldq;
  s_q[t] = qrow[t];
  float4 qf[4];
  if (RPE) {
    const float* qpr = qp + ((size_t)b * n + i) * 1024;
#pragma unroll
    for (int h = 0; h < 4; ++h) qf[h] = *reinterpret_cast<const float4*>(qpr + h * 256 + lane * 4);
  }
  __syncthreads();
  const float4 ql = *reinterpret_cast<const float4*>(&s_q[lane * 4]);  // lane covers head lane/16
  const float* Eb = RPE ? E + ((size_t)b * n + i) * (size_t)m * 256 : nullptr;
  const float* kb = k + (size_t)b * sk;
  // Reduction plan per key row: the four per-head partial dots of a lane are folded with a transpose-butterfly
  // (xor 32 keeps two heads per half, xor 16 one head per quarter, then 4 steps inside the 16-lane group): 7 shuffles
  // instead of 24.  After it, lane group g = lane>>4 holds head g -- the same group that owns head g of q.k.
  // 4 key rows per wave iteration: the 8 independent 16-byte loads (E row + k row each) are issued back to back so
  // that every wave keeps 8 KiB in flight instead of 2 KiB (the kernel is a pure HBM stream over E)
  for (int j0 = wave * 4; j0 < m; j0 += 16) {
    float4 kv[4], ev[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = min(j0 + u, m - 1);
      kv[u] = *reinterpret_cast<const float4*>(kb + (size_t)j * ldk + lane * 4);
      // E is streamed once per layer (2.5 GB per launch): non-temporal, so it does not evict the L2-resident k/v rows
      if (RPE) {
        const fx4 e = __builtin_nontemporal_load(reinterpret_cast<const fx4*>(Eb + (size_t)j * 256 + lane * 4));
        ev[u] = make_float4(e.x, e.y, e.z, e.w);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
      float se = (ql.x * kv[u].x + ql.y * kv[u].y) + (ql.z * kv[u].z + ql.w * kv[u].w);
      if (RPE) {
        float sp[4];
#pragma unroll
        for (int h = 0; h < 4; ++h)
          sp[h] = (qf[h].x * ev[u].x + qf[h].y * ev[u].y) + (qf[h].z * ev[u].z + qf[h].w * ev[u].w);
        const bool hi32 = lane & 32, hi16 = lane & 16;
        // keep heads {0,1} in lanes 0-31 and {2,3} in lanes 32-63
        const float s0 = hi32 ? sp[0] : sp[2], s1 = hi32 ? sp[1] : sp[3];  // what the partner half needs
        const float k0 = hi32 ? sp[2] : sp[0], k1 = hi32 ? sp[3] : sp[1];
        const float pa = k0 + xor32_f32(s0), pb = k1 + xor32_f32(s1);
        // keep the even head of the pair in lanes with bit4 = 0, the odd one in lanes with bit4 = 1
        const float send = hi16 ? pa : pb, keep = hi16 ? pb : pa;
        se += keep + xor16_f32(send);  // lane group (lane>>4) now carries head (lane>>4) of both terms
      }
      se = row16_sum_dpp(se);  // 4 DPP adds (row_ror 8/4/2/1) instead of 4 LDS-crossbar shuffles
      if ((lane & 15) == 0 && j < m) s_s[lane >> 4][j] = se * scale;
    }
  }
  __syncthreads();
  {  // softmax of head `wave` over m (F.softmax: exp(x - max) / sum)
    float mx = -INFINITY;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, s_s[wave][j]);
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int j = lane; j < m; j += 64) {
      const float e = expf(s_s[wave][j] - mx);
      s_s[wave][j] = e;
      sum += e;
    }
    sum = wave_sum_dpp(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < m; j += 64) s_s[wave][j] *= inv;
  }
  __syncthreads();
  const float* vb = v + (size_t)b * sv + t;
  const float* pr = s_s[t >> 6];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int j = 0;
  for (; j + 4 <= m; j += 4) {
    a0 = fmaf(pr[j], vb[(size_t)j * ldv], a0);
    a1 = fmaf(pr[j + 1], vb[(size_t)(j + 1) * ldv], a1);
    a2 = fmaf(pr[j + 2], vb[(size_t)(j + 2) * ldv], a2);
    a3 = fmaf(pr[j + 3], vb[(size_t)(j + 3) * ldv], a3);
  }
  for (; j < m; ++j) a0 = fmaf(pr[j], vb[(size_t)j * ldv], a0);
  out[(size_t)b * so + (size_t)i * ldo + t] = (a0 + a1) + (a2 + a3);
}

// Cross layers (no geometric term): NQ query tokens per workgroup.  The kernel is bound by the L2 reads of the cloud's k and v
// rows (2 x 197 KiB per workgroup): every k / v row load is shared by the NQ queries.
template <int NQ>
__global__ __launch_bounds__(256) void attention_mq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, float* __restrict__ out, int n, int m,
                                                           long ldq, long ldk, long ldv, long ldo, long sq, long sk, long sv,
                                                           long so, float scale) {
  __shared__ float s_s[NQ][4][AT_MAXM];
  const int b = blockIdx.y, i0 = blockIdx.x * NQ;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* qrow0 = q + (size_t)b * sq;
  float4 qv[NQ];  // lane covers head lane/16
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) qv[qi] = *reinterpret_cast<const float4*>(qrow0 + (size_t)min(i0 + qi, n - 1) * ldq + lane * 4);
  const float* kb = k + (size_t)b * sk;
  // (the k rows of the next step are requested before this step's dot products: the loop is bound by L2 latency, not by VALU)
  float4 kv[4], kn[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) kv[u] = *reinterpret_cast<const float4*>(kb + (size_t)min(wave * 4 + u, m - 1) * ldk + lane * 4);
  for (int j0 = wave * 4; j0 < m; j0 += 16) {
#pragma unroll
    for (int u = 0; u < 4; ++u) kn[u] = *reinterpret_cast<const float4*>(kb + (size_t)min(j0 + 16 + u, m - 1) * ldk + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u;
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        float sa = (qv[qi].x * kv[u].x + qv[qi].y * kv[u].y) + (qv[qi].z * kv[u].z + qv[qi].w * kv[u].w);
        sa = row16_sum_dpp(sa);
        if ((lane & 15) == 0 && j < m) s_s[qi][lane >> 4][j] = sa * scale;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) kv[u] = kn[u];
  }
  __syncthreads();
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) {  // softmax of head `wave` of query qi (F.softmax: exp(x - max) / sum)
    float* s = s_s[qi][wave];
    float mx = -INFINITY;
    for (int j = lane; j < m; j += 64) mx = fmaxf(mx, s[j]);
    mx = wave_max_dpp(mx);
    float sum = 0.f;
    for (int j = lane; j < m; j += 64) {
      const float e = expf(s[j] - mx);
      s[j] = e;
      sum += e;
    }
    sum = wave_sum_dpp(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < m; j += 64) s[j] *= inv;
  }
  __syncthreads();
  const float* vb = v + (size_t)b * sv + t;
  float acc[NQ][2];
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi) acc[qi][0] = acc[qi][1] = 0.f;
  int j = 0;
  for (; j + 8 <= m; j += 8) {  // eight v rows in flight; each accumulator sees its terms in the same order as in the 2-step loop
    float vv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) vv[u] = vb[(size_t)(j + u) * ldv];
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
#pragma unroll
      for (int qi = 0; qi < NQ; ++qi) {
        acc[qi][0] = fmaf(s_s[qi][t >> 6][j + u], vv[u], acc[qi][0]);
        acc[qi][1] = fmaf(s_s[qi][t >> 6][j + u + 1], vv[u + 1], acc[qi][1]);
      }
    }
  }
  for (; j + 2 <= m; j += 2) {
    const float v0 = vb[(size_t)j * ldv], v1 = vb[(size_t)(j + 1) * ldv];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) {
      acc[qi][0] = fmaf(s_s[qi][t >> 6][j], v0, acc[qi][0]);
      acc[qi][1] = fmaf(s_s[qi][t >> 6][j + 1], v1, acc[qi][1]);
    }
  }
  for (; j < m; ++j) {
    const float vv = vb[(size_t)j * ldv];
#pragma unroll
    for (int qi = 0; qi < NQ; ++qi) acc[qi][0] = fmaf(s_s[qi][t >> 6][j], vv, acc[qi][0]);
  }
#pragma unroll
  for (int qi = 0; qi < NQ; ++qi)
    if (i0 + qi < n) out[(size_t)b * so + (size_t)(i0 + qi) * ldo + t] = acc[qi][0] + acc[qi][1];
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
  dim3 grid(n, B);
  if (E)
    hipLaunchKernelGGL(attention_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, q, k, v, qp, E, out, n, m, ldq,
                       ldk, ldv, ldo, sq, sk, sv, so, scale);
  else
    hipLaunchKernelGGL(attention_mq_kernel<4>, dim3((n + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, q, k, v, out, n, m, ldq,
                       ldk, ldv, ldo, sq, sk, sv, so, scale);
  SAM6D_LAUNCH_CHECK("attention");
}

// =========================================================================================================
// Stand-alone pieces of the attention modules for callers that use a sub-module directly (the drop-in forwards of
// pem/transformer.py; the fused kernels above / in rpe.hip / xattn.hip never materialise the probabilities):
//   scaled_softmax:  out[r][0..n) = softmax_m((a[r][m] + b[r][m]) * scale)     (MultiHeadAttention.forward, PEM/model/transformer.py:134-143;
//                    with b = the q . proj_p(E) term: RPEMultiHeadAttention.forward :404-412)
//   sinusoid_embed:  out[i][2 j], out[i][2 j + 1] = sin(x_i w_j), cos(x_i w_j)   (SinusoidalPositionalEmbedding.forward :269-285)
// =========================================================================================================
__global__ __launch_bounds__(256) void scaled_softmax_kernel(const float* __restrict__ a, const float* __restrict__ b, float scale,
                                                             long rows, int n, long lda, long ldb, float* __restrict__ out, long ldo) {
  const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* ar = a + r * lda;
  const float* br = b ? b + r * ldb : nullptr;
  float mx = -INFINITY;
  for (int m = lane; m < n; m += 64) mx = fmaxf(mx, (ar[m] + (br ? br[m] : 0.f)) * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int m = lane; m < n; m += 64) sum += expf((ar[m] + (br ? br[m] : 0.f)) * scale - mx);
  sum = wave_sum(sum);
  float* o = out + r * ldo;
  for (int m = lane; m < n; m += 64) o[m] = expf((ar[m] + (br ? br[m] : 0.f)) * scale - mx) / sum;
}

extern "C" int sam6d_scaled_softmax(const float* a, const float* b, float scale, long rows, int n, long lda, long ldb, float* out,
                                    long ldo, void* stream) {
  SAM6D_REQUIRE(a && out && rows >= 0 && n > 0 && lda >= n && ldo >= n && (!b || ldb >= n), "scaled_softmax: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(scaled_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, a, b, scale, rows, n,
                     lda, ldb, out, ldo);
  SAM6D_LAUNCH_CHECK("scaled_softmax");
}

__global__ __launch_bounds__(256) void sinusoid_embed_kernel(const float* __restrict__ x, long n, const float* __restrict__ div_term,
                                                             int half, float* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n * half) return;
  const long i = e / half;
  const int j = (int)(e - i * half);
  const float w = x[i] * div_term[j];  // the fp32 product the reference forms (transformer.py:278)
  // sin / cos of that fp32 value in double, rounded once: the device's fp32 sincosf is only good to ~0.3 ulp OF THE ARGUMENT (2e-5 at
  // w ~ 866, the bg point's distance index), torch's CPU sin / cos to ~1 ulp of the result
  double sn, cs;
  sincos((double)w, &sn, &cs);
  out[(i * half + j) * 2] = (float)sn;
  out[(i * half + j) * 2 + 1] = (float)cs;
}

extern "C" int sam6d_sinusoid_embed(const float* x, long n, const float* div_term, int d_model, float* out, void* stream) {
  SAM6D_REQUIRE(x && div_term && out && n >= 0 && d_model > 0 && (d_model & 1) == 0, "sinusoid_embed: bad arguments (even d_model)");
  if (n == 0) return 0;
  const int half = d_model / 2;
  hipLaunchKernelGGL(sinusoid_embed_kernel, dim3((unsigned)((n * half + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, n, div_term,
                     half, out);
  SAM6D_LAUNCH_CHECK("sinusoid_embed");
}
