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
// the d part is a 32-term dot per (pair, head); the a part is a (32 keys x 3) x 32 x 256 contraction per key tile on
// v_mfma_f32_32x32x16_f16 (fp16 hi/lo split, 3 products), followed by max over k, the 4 head dots on packed-fp32 VALU and a
// transposing DPP reduction over the 32 channel lanes.  Pairs outside [0, xmax] (the bg token: 2n-1 of n^2) take their
// bias-free E row from the compact buffer sam6d_geo_outliers filled, old-style (one wave per row).
// q.k comes in precomputed (one small batched GEMM), softmax happens here, P.V is a batched GEMM afterwards.
//
// One persistent workgroup per CU: 7 waves, wave w owns keys [32w, 32w+32) of the current query (n <= 224); the A_c image
// (36 KiB) stays in LDS; scores are double-buffered so that one barrier per query suffices.
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));

#define RP_ROW 144                 // bytes per coefficient row image: 32 hi halves | 32 lo halves | 16 B pad (geo.hip GC_ROW)
#define RP_WBYTES (256 * RP_ROW)   // 36 864
#define RP_MAXM 224
#define RP_K 32

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}

// T_0..T_31(u) by the three-term recurrence (fp32: 7e-7 worst case on the projected embedding, below the reference's own
// fp32 sin/cos argument rounding of 3e-6); `f(p, T_p)` consumes each order
template <typename F>
__device__ __forceinline__ void chebyshev32(float u, F&& f) {
  const float u2 = u + u;
  float t0 = 1.0f, t1 = u;
  f(0, t0);
  f(1, t1);
#pragma unroll
  for (int p = 2; p < RP_K; ++p) {
    const float tp = fmaf(u2, t1, -t0);
    t0 = t1;
    t1 = tp;
    f(p, tp);
  }
}

__global__ __launch_bounds__(448) void rpe_score_kernel(const float4* __restrict__ idx4, const int* __restrict__ pos,
                                                        const float* __restrict__ rows, const unsigned char* __restrict__ Wc,
                                                        const float* __restrict__ qp, const float* __restrict__ qd,
                                                        const float* __restrict__ Se, float* __restrict__ P, int n, int ldp,
                                                        long Q, float xmax, float scale) {
  __shared__ __attribute__((aligned(16))) unsigned char Aw[RP_WBYTES];
  __shared__ float sc[2][4][RP_MAXM];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int fr = lane & 31, fk = lane >> 5;
  for (int i = t; i < RP_WBYTES / 16; i += 448) reinterpret_cast<uint4*>(Aw)[i] = reinterpret_cast<const uint4*>(Wc)[i];
  __syncthreads();
  const float uscale = 2.0f / xmax;
  const int m0 = wave * 32;
  int buf = 0;
  for (long q = blockIdx.x; q < Q; q += gridDim.x, buf ^= 1) {
    float* scb = &sc[buf][0][0];
    if (m0 < n) {  // wave-uniform: this wave has keys
      const int key = m0 + fr;
      const bool valid = key < n;
      const long pair = q * n + min(key, n - 1);
      const float4 v = idx4[pair];
      const int ps = pos[pair];
      const bool listed = valid && ps >= 0;
      const unsigned listed_mask = (unsigned)(__ballot(listed && fk == 0) & 0xffffffffull);
      const float* qpq = qp + q * 1024;

      // ---- d part: heads 2 fk, 2 fk + 1 of this key; written together with the q.k score
      {
        const bool inside = v.x >= 0.f && v.x <= xmax;
        const float u = inside ? fmaf(v.x, uscale, -1.0f) : 0.0f;
        const float* qdq = qd + q * 128 + fk * 64;
        float s0 = 0.f, s1 = 0.f;
        float4 w0, w1;
        chebyshev32(u, [&](int p, float tp) {
          if ((p & 3) == 0) {
            w0 = *reinterpret_cast<const float4*>(qdq + p);
            w1 = *reinterpret_cast<const float4*>(qdq + 32 + p);
          }
          const float a = (p & 3) == 0 ? w0.x : (p & 3) == 1 ? w0.y : (p & 3) == 2 ? w0.z : w0.w;
          const float b = (p & 3) == 0 ? w1.x : (p & 3) == 1 ? w1.y : (p & 3) == 2 ? w1.z : w1.w;
          s0 = fmaf(a, tp, s0);
          s1 = fmaf(b, tp, s1);
        });
        if (valid && !listed) {
          const float* seq = Se + (q * 4 + 2 * fk) * ldp + key;
          scb[(2 * fk) * RP_MAXM + key] = s0 + seq[0];
          scb[(2 * fk + 1) * RP_MAXM + key] = s1 + seq[ldp];
        }
      }

      // ---- a part: basis rows of the three angular indices as MFMA A fragments (row = key fr, k = 16 ks + 8 fk + j)
      half8 ah[3][2], al[3][2];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float x = k == 0 ? v.y : k == 1 ? v.z : v.w;
        const bool inside = x >= 0.f && x <= xmax;
        const float u = inside ? fmaf(x, uscale, -1.0f) : 0.0f;
        float lo8[2][8], hi8[2][8];  // orders {8 fk .. 8 fk + 7} and {16 + 8 fk ..}
        chebyshev32(u, [&](int p, float tp) {
          const int ks = p >> 4, j = p & 7;
          if (((p >> 3) & 1) == 0) lo8[ks][j] = tp; else hi8[ks][j] = tp;
        });
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float f = fk ? hi8[ks][j] : lo8[ks][j];
            const _Float16 h = (_Float16)f;
            ah[k][ks][j] = h;
            al[k][ks][j] = (_Float16)(f - (float)h);
          }
      }

      // ---- contraction over the 8 channel blocks; s01/s23[r] = partial head dots of key row r over this lane's channels
      f2 s01[16], s23[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        s01[r] = f2{0.f, 0.f};
        s23[r] = f2{0.f, 0.f};
      }
#pragma unroll 1
      for (int cb = 0; cb < 8; ++cb) {
        const int c = cb * 32 + fr;
        const float un = 1.0f / 1024.0f;  // the coefficient image is scaled by 1024 (exact power of two)
        const f2 q01 = f2{qpq[c] * un, qpq[256 + c] * un};
        const f2 q23 = f2{qpq[512 + c] * un, qpq[768 + c] * un};
        f32x16 acc[3];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          const unsigned char* wrow = Aw + (size_t)c * RP_ROW + (ks * 16 + fk * 8) * 2;
          const half8 bh = *reinterpret_cast<const half8*>(wrow);
          const half8 bl = *reinterpret_cast<const half8*>(wrow + 64);
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[k][ks], bh, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k][ks], bl, acc[k], 0, 0, 0);
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k][ks], bh, acc[k], 0, 0, 0);
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float g = __builtin_fmaxf(__builtin_fmaxf(acc[0][r], acc[1][r]), acc[2][r]);
          const f2 gg = f2{g, g};
          s01[r] = __builtin_elementwise_fma(gg, q01, s01[r]);
          s23[r] = __builtin_elementwise_fma(gg, q23, s23[r]);
        }
      }

      // ---- transposing reduction over the 32 channel lanes of each half: 64 values -> 2 per lane
      // value index = h * 16 + r; stage s pairs (2i, 2i+1) and keeps the one selected by a lane bit, so after the five
      // stages lane bits (b4 b3 b2 b1 b0) hold  r = b4 + 2 b3 + 4 b2 + 8 b1,  h = b0 + 2 i  (i = 0, 1)
      float V[64];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        V[r] = s01[r].x;
        V[16 + r] = s01[r].y;
        V[32 + r] = s23[r].x;
        V[48 + r] = s23[r].y;
      }
      float W1[32];
#pragma unroll
      for (int i = 0; i < 32; ++i) {  // bit 4: v_permlane16_swap exchanges the odd 16-lane rows of a with the even rows of b
        const u2 sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(V[2 * i]), __float_as_uint(V[2 * i + 1]), false, false);
        W1[i] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
      const bool b3 = lane & 8, b2 = lane & 4, b1 = lane & 2, b0 = lane & 1;
      float W2[16], W3[8], W4[4], Z[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float keep = b3 ? W1[2 * i + 1] : W1[2 * i], send = b3 ? W1[2 * i] : W1[2 * i + 1];
        W2[i] = keep + dpp_mov<0x140>(send);  // row_mirror: lane i <-> 15 - i
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float keep = b2 ? W2[2 * i + 1] : W2[2 * i], send = b2 ? W2[2 * i] : W2[2 * i + 1];
        W3[i] = keep + dpp_mov<0x141>(send);  // row_half_mirror: lane i <-> 7 - i
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float keep = b1 ? W3[2 * i + 1] : W3[2 * i], send = b1 ? W3[2 * i] : W3[2 * i + 1];
        W4[i] = keep + dpp_mov<0x4E>(send);  // quad_perm [2,3,0,1]
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float keep = b0 ? W4[2 * i + 1] : W4[2 * i], send = b0 ? W4[2 * i] : W4[2 * i + 1];
        Z[i] = keep + dpp_mov<0xB1>(send);  // quad_perm [1,0,3,2]
      }
      {
        const int r = ((lane >> 4) & 1) + 2 * (b3 ? 1 : 0) + 4 * (b2 ? 1 : 0) + 8 * (b1 ? 1 : 0);
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fk;
        const int h = b0 ? 1 : 0;
        if (m0 + row < n && !((listed_mask >> row) & 1u)) {
          scb[h * RP_MAXM + m0 + row] += Z[0];
          scb[(h + 2) * RP_MAXM + m0 + row] += Z[1];
        }
      }

      // ---- listed keys: the whole wave dots the stored embedding row with the 4 folded queries
      if (listed_mask) {  // wave-uniform
        float4 qf[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) qf[h] = *reinterpret_cast<const float4*>(qpq + h * 256 + lane * 4);
        unsigned mk = listed_mask;
        while (mk) {
          const int j = __builtin_ctz(mk);
          mk &= mk - 1;
          const int pj = __shfl(ps, j, 64);
          const float4 e = *reinterpret_cast<const float4*>(rows + (size_t)pj * 256 + lane * 4);
          float sp[4];
#pragma unroll
          for (int h = 0; h < 4; ++h) sp[h] = wave_sum((qf[h].x * e.x + qf[h].y * e.y) + (qf[h].z * e.z + qf[h].w * e.w));
          if (lane < 4) {
            const float mine = lane == 0 ? sp[0] : lane == 1 ? sp[1] : lane == 2 ? sp[2] : sp[3];
            scb[lane * RP_MAXM + m0 + j] = mine + Se[(q * 4 + lane) * ldp + m0 + j];
          }
        }
      }
    }
    __syncthreads();  // all scores of query q are in sc[buf]
    if (wave < 4) {   // softmax of head `wave` (F.softmax: exp(x - max) / sum), probabilities to P[q][head][:]
      const float* s = &sc[buf][wave][0];
      float mx = -INFINITY;
      for (int j = lane; j < n; j += 64) mx = fmaxf(mx, s[j] * scale);
      mx = wave_max(mx);
      float ev[4];
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        ev[u] = j < n ? expf(s[j] * scale - mx) : 0.f;
        sum += ev[u];
      }
      sum = wave_sum(sum);
      const float inv = 1.0f / sum;
      float* pr = P + (q * 4 + wave) * ldp;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = lane + 64 * u;
        if (j < n) pr[j] = ev[u] * inv;
      }
    }
  }
}

extern "C" int sam6d_rpe_scores(const float* idx_ws, const int* pos_ws, const float* rows, const void* wa_cheb, float xmax,
                                const float* qp, const float* qd, const float* qk, float* P, long Q, int n, int ldp,
                                void* stream) {
  SAM6D_REQUIRE(idx_ws && pos_ws && rows && wa_cheb && qp && qd && qk && P, "rpe_scores: null pointer");
  SAM6D_REQUIRE(Q >= 0 && n > 0 && n <= RP_MAXM && ldp >= n, "rpe_scores: need 0 < n <= %d and ldp >= n (n = %d, ldp = %d)",
                RP_MAXM, n, ldp);
  SAM6D_REQUIRE(xmax > 0.f, "rpe_scores: xmax must be positive");
  SAM6D_REQUIRE((((size_t)idx_ws | (size_t)wa_cheb | (size_t)qp | (size_t)qd | (size_t)rows) & 15) == 0,
                "rpe_scores: idx_ws / wa_cheb / qp / qd / rows must be 16-byte aligned");
  if (Q == 0) return 0;
  static int n_cu = 0;
  if (n_cu == 0) {
    int dev = 0, cu = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess || cu <= 0) {
      sam6d_set_error("rpe_scores: cannot query the device: %s", hipGetErrorString(e));
      return e != hipSuccess ? (int)e : SAM6D_EINVAL;
    }
    n_cu = cu;
  }
  const float scale = 0.125f;  // 1/sqrt(64): d_model 256, 4 heads (coarse_point_matching.py:24, fine_point_matching.py:31)
  hipLaunchKernelGGL(rpe_score_kernel, dim3((unsigned)(Q < n_cu ? Q : n_cu)), dim3(448), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(idx_ws), pos_ws, rows, reinterpret_cast<const unsigned char*>(wa_cheb), qp,
                     qd, qk, P, n, ldp, Q, xmax, scale);
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
