// fp32 "NT" GEMM on the CDNA4 matrix cores:  C = act(((A . W^T) / divisor) * colscale + bias) + residual
//   A (M,K) row-major (lda), W (N,K) row-major (ldw, torch nn.Linear weight layout), C (M,N) (ldc), optional batch.
// Serves every dense projection of the matching path (nn.Linear / 1x1 conv call sites: PEM/model/transformer.py:127-129,
// 186-188, 390-393, 548-550; PEM/model/coarse_point_matching.py:35-38; PEM/model/fine_point_matching.py:47-51) and the
// feature-similarity contraction (PEM/utils/model_utils.py:144).
//
// v_mfma_f32_32x32x2_f32: exact f32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD.  Block tile 128x128x16, 4 waves in a
// 2x2 grid, each wave a 64x64 sub-tile = 2x2 MFMA tiles (64 accumulator registers).  Operands are staged in LDS with
// an odd row stride (17 dwords) so the ds_read_b32 fragment reads (32 different rows per lane group) are
// conflict-free; the MFMA issue time (64 cycles each) dominates, several blocks per CU hide the staging.
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GM_BM 128
#define GM_BN 128
#define GM_BK 16
#define GM_LD 17

__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const float* __restrict__ A, const float* __restrict__ W,
                                                         const float* __restrict__ bias,
                                                         const float* __restrict__ colscale,
                                                         const float* __restrict__ residual, float* __restrict__ C,
                                                         int M, int N, int K, long lda, long ldw, long ldc, long ldr,
                                                         long sA, long sW, long sC, long sR, float divisor, int act) {
  __shared__ float As[GM_BM * GM_LD];
  __shared__ float Bs[GM_BN * GM_LD];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int bz = blockIdx.z;
  A += (size_t)bz * sA;
  W += (size_t)bz * sW;
  C += (size_t)bz * sC;
  if (residual) residual += (size_t)bz * sR;
  const int m0 = blockIdx.x * GM_BM, n0 = blockIdx.y * GM_BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging map: 128 rows x 16 k = 512 float4; thread t handles rows (t>>2) and (t>>2)+64, k4 = (t&3)*4
  const int sr = t >> 2, sk = (t & 3) * 4;
  const int ar0 = min(m0 + sr, M - 1), ar1 = min(m0 + sr + 64, M - 1);
  const int br0 = min(n0 + sr, N - 1), br1 = min(n0 + sr + 64, N - 1);
  const float* a0p = A + (size_t)ar0 * lda + sk;
  const float* a1p = A + (size_t)ar1 * lda + sk;
  const float* b0p = W + (size_t)br0 * ldw + sk;
  const float* b1p = W + (size_t)br1 * ldw + sk;
  const bool vec = ((lda & 3) == 0) && ((ldw & 3) == 0) && ((((size_t)A | (size_t)W) & 15) == 0);

  const int fr = lane & 31, fk = lane >> 5;
  for (int k0 = 0; k0 < K; k0 += GM_BK) {
    float4 va0, va1, vb0, vb1;
    if (vec && k0 + GM_BK <= K) {
      va0 = *reinterpret_cast<const float4*>(a0p + k0);
      va1 = *reinterpret_cast<const float4*>(a1p + k0);
      vb0 = *reinterpret_cast<const float4*>(b0p + k0);
      vb1 = *reinterpret_cast<const float4*>(b1p + k0);
    } else {
      float ta[4], tb[4], tc[4], td[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const bool ok = (k0 + sk + u) < K;
        ta[u] = ok ? a0p[k0 + u] : 0.f;
        tb[u] = ok ? a1p[k0 + u] : 0.f;
        tc[u] = ok ? b0p[k0 + u] : 0.f;
        td[u] = ok ? b1p[k0 + u] : 0.f;
      }
      va0 = make_float4(ta[0], ta[1], ta[2], ta[3]);
      va1 = make_float4(tb[0], tb[1], tb[2], tb[3]);
      vb0 = make_float4(tc[0], tc[1], tc[2], tc[3]);
      vb1 = make_float4(td[0], td[1], td[2], td[3]);
    }
    __syncthreads();  // previous tile fully consumed
    float* pa0 = As + sr * GM_LD + sk;
    float* pa1 = As + (sr + 64) * GM_LD + sk;
    float* pb0 = Bs + sr * GM_LD + sk;
    float* pb1 = Bs + (sr + 64) * GM_LD + sk;
    pa0[0] = va0.x; pa0[1] = va0.y; pa0[2] = va0.z; pa0[3] = va0.w;
    pa1[0] = va1.x; pa1[1] = va1.y; pa1[2] = va1.z; pa1[3] = va1.w;
    pb0[0] = vb0.x; pb0[1] = vb0.y; pb0[2] = vb0.z; pb0[3] = vb0.w;
    pb1[0] = vb1.x; pb1[1] = vb1.y; pb1[2] = vb1.z; pb1[3] = vb1.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < GM_BK; kk += 2) {
      const float a0 = As[(wm + fr) * GM_LD + kk + fk];
      const float a1 = As[(wm + 32 + fr) * GM_LD + kk + fk];
      const float b0 = Bs[(wn + fr) * GM_LD + kk + fk];
      const float b1 = Bs[(wn + 32 + fr) * GM_LD + kk + fk];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }
  // epilogue: C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 32 + fr;
      if (col >= N) continue;
      const float bv = bias ? bias[col] : 0.f;
      const float cs = colscale ? colscale[col] : 1.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fk;
        if (row < M) {
          float v = acc[i][j][r];
          if (divisor != 1.0f) v = v / divisor;
          v = colscale ? fmaf(v, cs, bv) : v + bv;
          if (act == 1) v = v > 0.f ? v : 0.f;
          if (residual) v += residual[(size_t)row * ldr + col];
          C[(size_t)row * ldc + col] = v;
        }
      }
    }
}

extern "C" int sam6d_gemm_nt(const float* A, const float* W, const float* bias, const float* colscale,
                             const float* residual, float* C, int M, int N, int K, long lda, long ldw, long ldc, long ldr,
                             int batch, long sA, long sW, long sC, long sR, float divisor, int act, void* stream) {
  SAM6D_REQUIRE(A && W && C, "gemm_nt: null pointer");
  SAM6D_REQUIRE(M >= 0 && N >= 0 && K > 0 && batch >= 0, "gemm_nt: bad sizes M=%d N=%d K=%d batch=%d", M, N, K, batch);
  SAM6D_REQUIRE(lda >= K && ldw >= K && ldc >= N, "gemm_nt: leading dimension smaller than the row length");
  SAM6D_REQUIRE(act == 0 || act == 1, "gemm_nt: act must be 0 (none) or 1 (ReLU)");
  SAM6D_REQUIRE(batch <= 65535, "gemm_nt: batch must be <= 65535");
  if (M == 0 || N == 0 || batch == 0) return 0;
  dim3 grid(cdiv(M, GM_BM), cdiv(N, GM_BN), batch);  // M tiles on x (2^31 limit): the PE MLP has 8.4 M rows
  SAM6D_REQUIRE(grid.y <= 65535, "gemm_nt: N too large for one launch (%d)", N);
  hipLaunchKernelGGL(gemm_nt_kernel, grid, dim3(256), 0, (hipStream_t)stream, A, W, bias, colscale, residual, C, M, N, K,
                     lda, ldw, ldc, ldr, sA, sW, sC, sR, divisor, act);
  SAM6D_LAUNCH_CHECK("gemm_nt");
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
  const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
  const float dx = v.x - mean, dy = v.y - mean, dz = v.z - mean, dw = v.w - mean;
  const float var = wave_sum((dx * dx + dy * dy) + (dz * dz + dw * dw)) * (1.0f / 256.0f);
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
