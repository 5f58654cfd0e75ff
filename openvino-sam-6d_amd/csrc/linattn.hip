// Focused linear attention of the dense (2048-token) lift, kv path (PEM/model/transformer.py:532-578):
//   phi(x) = x3 / |x3| * |x1|,  x1 = (relu(x) + 1e-6) / softplus(scale),  x3 = x1^3     (norms over all 256 channels)
//   per head (c = d = 64):  z_i = 1 / (phi(q)_i . sum_j phi(k)_j + 1e-6)
//                           kv  = sum_j phi(k)_j^T v_j                       (64 x 64)
//                           x_i = (phi(q)_i . kv) * z_i
// Three small kernels around the GEMM: focus_k (phi in place), kv_reduce (kv^T + key sums per (b,h)), focus_q (phi and
// the z scaling folded into the query rows so the final contraction is a plain batched GEMM with W = kv^T).
// All HBM-bound row work: one wave per 256-channel row, 16-byte lanes.
#include "common.h"
#include "../../include/sam6d_hip.h"

__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }  // nn.Softplus(beta=1, threshold=20)

__device__ __forceinline__ float4 focus_row(float4 x, const float4 sp) {
  // relu + 1e-6, / softplus(scale)
  float a[4] = {x.x, x.y, x.z, x.w};
  const float s[4] = {sp.x, sp.y, sp.z, sp.w};
  float n1 = 0.f, n3 = 0.f;
  float c[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    float v = (a[u] > 0.f ? a[u] : 0.f) + 1e-6f;
    v = v / s[u];
    a[u] = v;
    n1 += v * v;
    c[u] = (v * v) * v;
    n3 += c[u] * c[u];
  }
  n1 = sqrtf(wave_sum_dpp(n1));
  n3 = sqrtf(wave_sum_dpp(n3));
  return make_float4((c[0] / n3) * n1, (c[1] / n3) * n1, (c[2] / n3) * n1, (c[3] / n3) * n1);
}

// phi() in place on rows of 256 floats (row stride ld)
__global__ __launch_bounds__(256) void focus_k_kernel(float* __restrict__ x, const float* __restrict__ scale, long rows,
                                                      long ld) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 sc = *reinterpret_cast<const float4*>(scale + lane * 4);
  const float4 sp = make_float4(softplus_f(sc.x), softplus_f(sc.y), softplus_f(sc.z), softplus_f(sc.w));
  float4* p = reinterpret_cast<float4*>(x + row * ld + lane * 4);
  *p = focus_row(*p, sp);
}

// per (b,h): kvT[d][c] = sum_j k[j,h*64+c] v[j,h*64+d], ksum[c] = sum_j k[j,h*64+c].   256 threads, thread owns
// output (d = t>>2, c in [(t&3)*16, +16)); k/v rows staged through LDS in tiles of 32 keys.
__global__ __launch_bounds__(256) void kv_reduce_kernel(const float* __restrict__ k, const float* __restrict__ v, int J,
                                                        long ldk, long ldv, long sk, long sv, float* __restrict__ kvT,
                                                        float* __restrict__ ksum) {
  __shared__ float ks[32][64];
  __shared__ float vs[32][65];
  const int b = blockIdx.y, h = blockIdx.x, t = threadIdx.x;
  const float* kb = k + (size_t)b * sk + h * 64;
  const float* vb = v + (size_t)b * sv + h * 64;
  const int d = t >> 2, c0 = (t & 3) * 16;
  float acc[16];
#pragma unroll
  for (int u = 0; u < 16; ++u) acc[u] = 0.f;
  float ksacc = 0.f;  // threads 0..63 accumulate ksum[c = t]
  for (int j0 = 0; j0 < J; j0 += 32) {
    __syncthreads();
    for (int e = t; e < 32 * 64; e += 256) {
      const int jj = e >> 6, cc = e & 63;
      const bool ok = (j0 + jj) < J;
      ks[jj][cc] = ok ? kb[(size_t)(j0 + jj) * ldk + cc] : 0.f;
      vs[jj][cc] = ok ? vb[(size_t)(j0 + jj) * ldv + cc] : 0.f;
    }
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < 32; ++jj) {
      const float vv = vs[jj][d];
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u] = fmaf(ks[jj][c0 + u], vv, acc[u]);
      if (t < 64) ksacc += ks[jj][t];
    }
  }
  float* o = kvT + ((size_t)b * 4 + h) * 4096 + d * 64 + c0;
#pragma unroll
  for (int u = 0; u < 16; ++u) o[u] = acc[u];
  if (t < 64) ksum[((size_t)b * 4 + h) * 64 + t] = ksacc;
}

// q rows: phi() then scale head h by z = 1 / (phi(q)_h . ksum_h + 1e-6)
__global__ __launch_bounds__(256) void focus_q_kernel(float* __restrict__ x, const float* __restrict__ scale,
                                                      const float* __restrict__ ksum, long rows_per_b, long rows, long ld) {
  const int lane = threadIdx.x & 63;
  // softplus(scale) costs as much as the rest of a row: each wave computes it once and walks rows with a grid stride
  const float4 sc = *reinterpret_cast<const float4*>(scale + lane * 4);
  const float4 sp = make_float4(softplus_f(sc.x), softplus_f(sc.y), softplus_f(sc.z), softplus_f(sc.w));
  const long stride = (long)gridDim.x * 4;
  for (long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += stride) {
    const long b = row / rows_per_b;
    float4* p = reinterpret_cast<float4*>(x + row * ld + lane * 4);
    float4 q = focus_row(*p, sp);
    const float4 ksv = *reinterpret_cast<const float4*>(ksum + b * 256 + lane * 4);  // [b][h][64] == [b][256]
    float dot = (q.x * ksv.x + q.y * ksv.y) + (q.z * ksv.z + q.w * ksv.w);
    dot = row16_sum_dpp(dot);  // 16-lane group == one head
    const float z = 1.0f / (dot + 1e-6f);
    *p = make_float4(q.x * z, q.y * z, q.z * z, q.w * z);
  }
}

extern "C" int sam6d_linattn_focus_k(float* k, const float* scale, long rows, long ld, void* stream) {
  SAM6D_REQUIRE(k && scale && rows >= 0 && ld >= 256 && (ld & 3) == 0, "linattn_focus_k: bad arguments");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(focus_k_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, k, scale, rows, ld);
  SAM6D_LAUNCH_CHECK("linattn_focus_k");
}

extern "C" int sam6d_linattn_kv(const float* k, const float* v, int B, int J, long ldk, long ldv, long sk, long sv,
                                float* kvT, float* ksum, void* stream) {
  SAM6D_REQUIRE(k && v && kvT && ksum && B >= 0 && J > 0 && B <= 65535, "linattn_kv: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(kv_reduce_kernel, dim3(4, B), dim3(256), 0, (hipStream_t)stream, k, v, J, ldk, ldv, sk, sv, kvT, ksum);
  SAM6D_LAUNCH_CHECK("linattn_kv");
}

extern "C" int sam6d_linattn_focus_q(float* q, const float* scale, const float* ksum, int B, long rows_per_b, long ld,
                                     void* stream) {
  SAM6D_REQUIRE(q && scale && ksum && B >= 0 && rows_per_b > 0 && ld >= 256 && (ld & 3) == 0, "linattn_focus_q: bad arguments");
  if (B == 0) return 0;
  const long rows = (long)B * rows_per_b;
  long blocks = (rows + 3) / 4;
  if (blocks > 4096) blocks = 4096;  // 16 workgroups per CU, 8 rows per wave at the dense size
  hipLaunchKernelGGL(focus_q_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, q, scale, ksum, rows_per_b, rows,
                     ld);
  SAM6D_LAUNCH_CHECK("linattn_focus_q");
}
