// Pose solvers of the matching path for gfx950: soft assignment, weighted hypothesis sampling, batched 3-point and
// N-point weighted Procrustes (3x3 SVD), hypothesis selection / scoring.
//   compute_coarse_Rt      PEM/utils/model_utils.py:204-275  (+ weighted_sampling_onnx_compatible :277-305)
//   compute_fine_Rt        PEM/utils/model_utils.py:308-341
//   weighted_procrustes    PEM/utils/model_utils.py:343-436  (torch.svd / torch.det :469-481, 513-526)
// HBM/latency-bound vector work: coalesced row passes, wave butterflies, LDS-staged point sets; no MFMA.
#include "common.h"
#include "../../include/sam6d_hip.h"

#include <math.h>

// v ** 1.5 of the sampling weights (model_utils.py:238) as v * sqrtf(v): two correctly rounded operations (<= 1 ulp of the real value,
// like the ~100-instruction generic powf and like torch's own pow kernel); 38 416 of them per proposal sat in one workgroup's epilogue.
// The sampled indices of tests/golden/coarse_rt.npz stay bit-exact; on config 2 every index remains a first-(cum >= u) index of the
// oracle's cumulative weights within 1e-6 (test_config2_full_batch_vs_oracle).
__device__ __forceinline__ float sa_pow15(float v) { return v * sqrtf(v); }

// =========================================================================================================
// Soft assignment  S = softmax(att, dim=2) * softmax(att, dim=1)   (model_utils.py:229-233, 320-324)
// att (B, R, C), row 0 / column 0 = background token.
// =========================================================================================================
#define SA_RPL 36
__global__ __launch_bounds__(256) void sa_row_stats_kernel(const float* __restrict__ att, int C, long rows,
                                                           float* __restrict__ rmax, float* __restrict__ rsum) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* a = att + row * C;
  float mx = -INFINITY, s = 0.f;
  if (C <= 64 * SA_RPL) {
    // the row (<= 2304 floats) is read ONCE into registers, all loads in flight together; max and sum-exp then run from
    // registers in the same per-lane order as the two-loop form below (identical bits)
    float v[SA_RPL];
#pragma unroll
    for (int k = 0; k < SA_RPL; ++k) {
      const int c = lane + 64 * k;
      v[k] = (c < C) ? a[c] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < SA_RPL; ++k) mx = fmaxf(mx, v[k]);
    mx = wave_max(mx);
#pragma unroll
    for (int k = 0; k < SA_RPL; ++k)
      if (lane + 64 * k < C) s += expf(v[k] - mx);
  } else {
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, a[c]);
    mx = wave_max(mx);
    for (int c = lane; c < C; c += 64) s += expf(a[c] - mx);
  }
  s = wave_sum(s);
  if (lane == 0) {
    rmax[row] = mx;
    rsum[row] = s;
  }
}

// Column statistics: the R rows are cut into SA_RS slices so that (C/256) x B x SA_RS workgroups stream the matrix
// (a single thread per column walking all 2049 rows left 7/8 of the chip idle); a second tiny kernel merges the
// per-slice (max, sum-exp) pairs:  m = max_s m_s,  sum = sum_s sum_s * exp(m_s - m).
// Small matrices (the 197 x 197 coarse attention) use ONE slice: the column sum is then the plain sequential fp32 sum
// in row order, which is what the reference's softmax(dim=1) computes -- the sampled hypothesis indices downstream
// stay bit-exact with the reference when the same attention matrix is injected.
#define SA_RS_MAX 16
__global__ __launch_bounds__(256) void sa_col_stats_part_kernel(const float* __restrict__ att, int R, int C, int SA_RS,
                                                                float* __restrict__ pmax, float* __restrict__ psum) {
  const int b = blockIdx.y, rs = blockIdx.z;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const int per = (R + SA_RS - 1) / SA_RS;
  const int r0 = rs * per, r1 = min(R, r0 + per);
  const float* a = att + (size_t)b * R * C + c;
  // (unrolled: the strided loads of 8 rows are in flight together; the fp32 sum itself stays sequential in row order)
  float mx = -INFINITY;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) mx = fmaxf(mx, a[(size_t)r * C]);
  float s = 0.f;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) s += expf(a[(size_t)r * C] - mx);
  pmax[((size_t)b * SA_RS + rs) * C + c] = mx;
  psum[((size_t)b * SA_RS + rs) * C + c] = s;
}

__global__ __launch_bounds__(256) void sa_col_stats_merge_kernel(const float* __restrict__ pmax, const float* __restrict__ psum,
                                                                 int C, int SA_RS, float* __restrict__ cmax,
                                                                 float* __restrict__ csum) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  if (SA_RS == 1) {  // single slice: pass the sequential sum through untouched
    cmax[(size_t)b * C + c] = pmax[(size_t)b * C + c];
    csum[(size_t)b * C + c] = psum[(size_t)b * C + c];
    return;
  }
  float mx = -INFINITY;
  for (int rs = 0; rs < SA_RS; ++rs) mx = fmaxf(mx, pmax[((size_t)b * SA_RS + rs) * C + c]);
  float s = 0.f;
  for (int rs = 0; rs < SA_RS; ++rs) {
    const float m = pmax[((size_t)b * SA_RS + rs) * C + c];
    if (m != -INFINITY) s += psum[((size_t)b * SA_RS + rs) * C + c] * expf(m - mx);  // empty slices hold (-inf, 0)
  }
  cmax[(size_t)b * C + c] = mx;
  csum[(size_t)b * C + c] = s;
}

__device__ __forceinline__ float sa_value(float a, float rm, float rs, float cm, float cs) {
  return (expf(a - rm) / rs) * (expf(a - cm) / cs);
}
// The same product with the hardware exponential and reciprocal (v_exp_f32 / v_rcp_f32: ~1e-6 relative) -- 10 VALU
// instructions instead of ~45.  Used for the LARGE (fine-stage, 2049 x 2049) matrices only, where the label / assignment
// passes were bound by this arithmetic rather than by the 537 MB they stream; the 197 x 197 coarse matrix keeps the exact
// form (its weights feed the hypothesis sampling, which is pinned index for index against the reference).
template <bool FAST>
__device__ __forceinline__ float sa_val(float a, float rm, float rs, float cm, float cs) {
  if (FAST) return (__expf(a - rm) * __frcp_rn(rs)) * (__expf(a - cm) * __frcp_rn(cs));
  return sa_value(a, rm, rs, cm, cs);
}

// label1[b, r-1] = argmax_c S[b, r, c] (first maximum), r = 1..R-1  -- one wave per row
template <bool FAST>
__global__ __launch_bounds__(256) void sa_row_labels_kernel(const float* __restrict__ att, int R, int C, long rows,
                                                            const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                            const float* __restrict__ cmax, const float* __restrict__ csum,
                                                            int* __restrict__ label1) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*(R-1)
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  const long b = w / (R - 1);
  const int r = (int)(w % (R - 1)) + 1;
  const float* a = att + ((size_t)b * R + r) * C;
  const float rm = rmax[b * R + r], rs = rsum[b * R + r];
  const float* cm = cmax + b * C;
  const float* cs = csum + b * C;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = lane; c < C; c += 64) {
    const float v = sa_val<FAST>(a[c], rm, rs, cm[c], cs[c]);
    if (v > best) {
      best = v;
      bi = c;
    }
  }
  wave_argmax_first(best, bi);
  if (lane == 0) label1[w] = (bi == 0x7fffffff) ? 0 : bi;
}

// label2[b, c-1] = argmax_r S[b, r, c] (first maximum), c = 1..C-1: row-sliced like the column statistics; the merge
// keeps the lowest row on ties, so the result equals the sequential scan's.
template <bool FAST>
__global__ __launch_bounds__(256) void sa_col_labels_part_kernel(const float* __restrict__ att, int R, int C, int SA_RS,
                                                                 const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                                 const float* __restrict__ cmax, const float* __restrict__ csum,
                                                                 float* __restrict__ pbest, int* __restrict__ pidx) {
  const int b = blockIdx.y, rs = blockIdx.z;
  const int c = blockIdx.x * 256 + threadIdx.x + 1;
  if (c >= C) return;
  const int per = (R + SA_RS - 1) / SA_RS;
  const int r0 = rs * per, r1 = min(R, r0 + per);
  const float* a = att + (size_t)b * R * C + c;
  const float cm = cmax[(size_t)b * C + c], cs = csum[(size_t)b * C + c];
  const float* rm = rmax + (size_t)b * R;
  const float* rsm = rsum + (size_t)b * R;
  float best = -INFINITY;
  int bi = 0x7fffffff;
#pragma unroll 8
  for (int r = r0; r < r1; ++r) {
    const float v = sa_val<FAST>(a[(size_t)r * C], rm[r], rsm[r], cm, cs);
    if (v > best) {
      best = v;
      bi = r;
    }
  }
  pbest[((size_t)b * SA_RS + rs) * C + c] = best;
  pidx[((size_t)b * SA_RS + rs) * C + c] = bi;
}

__global__ __launch_bounds__(256) void sa_col_labels_merge_kernel(const float* __restrict__ pbest, const int* __restrict__ pidx,
                                                                  int C, int SA_RS, int* __restrict__ label2) {
  const int b = blockIdx.y;
  const int c = blockIdx.x * 256 + threadIdx.x + 1;
  if (c >= C) return;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int rs = 0; rs < SA_RS; ++rs) {  // slices are in increasing row order: strict > keeps the first maximum
    const float v = pbest[((size_t)b * SA_RS + rs) * C + c];
    if (v > best) {
      best = v;
      bi = pidx[((size_t)b * SA_RS + rs) * C + c];
    }
  }
  label2[(size_t)b * (C - 1) + (c - 1)] = (bi == 0x7fffffff) ? 0 : bi;
}

extern "C" int sam6d_soft_assign(const float* att, int B, int R, int C, float* rmax, float* rsum, float* cmax, float* csum,
                                 int* label1, int* label2, float* ws, long ws_floats, void* stream) {
  SAM6D_REQUIRE(att && rmax && rsum && cmax && csum && label1 && label2 && ws, "soft_assign: null pointer");
  SAM6D_REQUIRE(B >= 0 && R >= 2 && C >= 2 && B <= 65535, "soft_assign: bad sizes");
  const int SA_RS = (R <= 256) ? 1 : SA_RS_MAX;
  SAM6D_REQUIRE(ws_floats >= 2L * B * SA_RS * C, "soft_assign: workspace needs 2*B*%d*C floats", SA_RS);
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  float* p0 = ws;
  float* p1 = ws + (size_t)B * SA_RS * C;
  const long rows = (long)B * R;
  hipLaunchKernelGGL(sa_row_stats_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, att, C, rows, rmax, rsum);
  hipLaunchKernelGGL(sa_col_stats_part_kernel, dim3(cdiv(C, 256), B, SA_RS), dim3(256), 0, s, att, R, C, SA_RS, p0, p1);
  hipLaunchKernelGGL(sa_col_stats_merge_kernel, dim3(cdiv(C, 256), B), dim3(256), 0, s, p0, p1, C, SA_RS, cmax, csum);
  const long lrows = (long)B * (R - 1);
  if (R > 256) {  // fine-stage sizes: hardware exp / rcp (see sa_val)
    hipLaunchKernelGGL(sa_row_labels_kernel<true>, dim3((unsigned)((lrows + 3) / 4)), dim3(256), 0, s, att, R, C, lrows, rmax, rsum,
                       cmax, csum, label1);
    hipLaunchKernelGGL(sa_col_labels_part_kernel<true>, dim3(cdiv(C - 1, 256), B, SA_RS), dim3(256), 0, s, att, R, C, SA_RS, rmax,
                       rsum, cmax, csum, p0, reinterpret_cast<int*>(p1));
  } else {
    hipLaunchKernelGGL(sa_row_labels_kernel<false>, dim3((unsigned)((lrows + 3) / 4)), dim3(256), 0, s, att, R, C, lrows, rmax, rsum,
                       cmax, csum, label1);
    hipLaunchKernelGGL(sa_col_labels_part_kernel<false>, dim3(cdiv(C - 1, 256), B, SA_RS), dim3(256), 0, s, att, R, C, SA_RS, rmax,
                       rsum, cmax, csum, p0, reinterpret_cast<int*>(p1));
  }
  hipLaunchKernelGGL(sa_col_labels_merge_kernel, dim3(cdiv(C - 1, 256), B), dim3(256), 0, s, p0, reinterpret_cast<const int*>(p1),
                     C, SA_RS, label2);
  SAM6D_LAUNCH_CHECK("soft_assign");
}

// coarse: weights[b, (r-1)*(C-1) + (c-1)] = (S * [label1>0] * [label2>0]) ^ 1.5 ; w1[b, r-1] = [label1 > 0]
__global__ __launch_bounds__(256) void coarse_weights_kernel(const float* __restrict__ att, int R, int C,
                                                             const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                             const float* __restrict__ cmax, const float* __restrict__ csum,
                                                             const int* __restrict__ label1, const int* __restrict__ label2,
                                                             float* __restrict__ weights, float* __restrict__ w1, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int n2 = C - 1, n1 = R - 1;
  const int c = (int)(e % n2) + 1;
  const long br = e / n2;
  const int r = (int)(br % n1) + 1;
  const long b = br / n1;
  const float a = att[((size_t)b * R + r) * C + c];
  float v = sa_value(a, rmax[b * R + r], rsum[b * R + r], cmax[b * C + c], csum[b * C + c]);
  const float f1 = label1[b * n1 + (r - 1)] > 0 ? 1.f : 0.f;
  const float f2 = label2[b * n2 + (c - 1)] > 0 ? 1.f : 0.f;
  v = (v * f1) * f2;
  weights[e] = sa_pow15(v);
  if (c == 1) w1[b * n1 + (r - 1)] = f1;
}

extern "C" int sam6d_coarse_weights(const float* att, int B, int R, int C, const float* rmax, const float* rsum,
                                    const float* cmax, const float* csum, const int* label1, const int* label2,
                                    float* weights, float* w1, void* stream) {
  SAM6D_REQUIRE(att && rmax && rsum && cmax && csum && label1 && label2 && weights && w1, "coarse_weights: null pointer");
  const long total = (long)B * (R - 1) * (C - 1);
  if (total == 0) return 0;
  hipLaunchKernelGGL(coarse_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, att, R,
                     C, rmax, rsum, cmax, csum, label1, label2, weights, w1, total);
  SAM6D_LAUNCH_CHECK("coarse_weights");
}

// The coarse (197 x 197) soft assignment and its weights in ONE launch: one workgroup of 1024 threads per proposal holds the
// attention matrix in LDS (155 KB) and runs the six passes of sam6d_soft_assign + sam6d_coarse_weights back to back, each with the
// arithmetic and the summation order of the kernel it replaces (row statistics: sa_row_stats_kernel's per-lane order + wave
// reductions; column sums: ONE thread per column in row order -- the reference's sequential softmax(dim=1) sum, which the bit-exact
// hypothesis indices rest on; the arg-max passes only compare per-element values, so a wave per column finds the same first maximum
// as the sequential scan).  Six launches (216 us, two of them a single thread per column walking global memory) + coarse_weights
// (62 us) on the critical path of the pose-solver phase.
__global__ __launch_bounds__(1024) void coarse_assign_kernel(const float* __restrict__ att, int R, int C, float* __restrict__ rmax,
                                                             float* __restrict__ rsum, float* __restrict__ cmax,
                                                             float* __restrict__ csum, int* __restrict__ label1,
                                                             int* __restrict__ label2, float* __restrict__ weights,
                                                             float* __restrict__ w1) {
  extern __shared__ __attribute__((aligned(16))) float cas[];
  float* m = cas;                 // [R][C]
  float* rm = m + (size_t)R * C;  // [R]
  float* rs = rm + R;
  float* cm = rs + R;             // [C]
  float* cs = cm + C;
  int* l1 = reinterpret_cast<int*>(cs + C);  // [R] (entry r: label of row r, r >= 1)
  int* l2 = l1 + R;                          // [C]
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* a = att + (size_t)b * R * C;
  for (int i = t; i < R * C; i += 1024) m[i] = a[i];
  __syncthreads();
  if (wave < 12) {
    // ---- row statistics (sa_row_stats_kernel, C <= 2304 form): waves 0..11
    for (int r = wave; r < R; r += 12) {
      const float* row = m + (size_t)r * C;
      float mx = -INFINITY, s = 0.f;
      for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
      mx = wave_max(mx);
      for (int c = lane; c < C; c += 64) s += expf(row[c] - mx);
      s = wave_sum(s);
      if (lane == 0) {
        rm[r] = mx;
        rs[r] = s;
      }
    }
  } else {
    // ---- column statistics (sa_col_stats_part_kernel with one slice), beside the row pass: one thread per column, sequential in row
    // order
    for (int c = t - 768; c < C; c += 256) {
      float mx = -INFINITY;
      for (int r = 0; r < R; ++r) mx = fmaxf(mx, m[(size_t)r * C + c]);
      float s = 0.f;
#pragma unroll 4
      for (int r = 0; r < R; ++r) s += expf(m[(size_t)r * C + c] - mx);
      cm[c] = mx;
      cs[c] = s;
    }
  }
  __syncthreads();
  // ---- S = softmax(att, 2) * softmax(att, 1), once, in place of the matrix (the three passes below read the same S values the
  // separate kernels recompute: sa_value is a pure function of the element and its four statistics)
  for (int i = t; i < R * C; i += 1024) {
    const int r = i / C, c = i - r * C;
    m[i] = sa_value(m[i], rm[r], rs[r], cm[c], cs[c]);
  }
  __syncthreads();
  // ---- row labels (sa_row_labels_kernel<false>): first maximum over the columns
  for (int r = 1 + wave; r < R; r += 16) {
    const float* row = m + (size_t)r * C;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
      const float v = row[c];
      if (v > best) {
        best = v;
        bi = c;
      }
    }
    wave_argmax_first(best, bi);
    if (lane == 0) l1[r] = (bi == 0x7fffffff) ? 0 : bi;
  }
  // ---- column labels: first maximum over the rows (a wave per column; ties -> the lower row, as the sequential scan keeps)
  for (int c = 1 + wave; c < C; c += 16) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int r = lane; r < R; r += 64) {
      const float v = m[(size_t)r * C + c];
      if (v > best) {
        best = v;
        bi = r;
      }
    }
    wave_argmax_first(best, bi);
    if (lane == 0) l2[c] = (bi == 0x7fffffff) ? 0 : bi;
  }
  __syncthreads();
  // ---- outputs: statistics, labels, weights (coarse_weights_kernel)
  for (int i = t; i < R; i += 1024) {
    rmax[(size_t)b * R + i] = rm[i];
    rsum[(size_t)b * R + i] = rs[i];
    if (i >= 1) {
      label1[(size_t)b * (R - 1) + i - 1] = l1[i];
      w1[(size_t)b * (R - 1) + i - 1] = l1[i] > 0 ? 1.f : 0.f;
    }
  }
  for (int i = t; i < C; i += 1024) {
    cmax[(size_t)b * C + i] = cm[i];
    csum[(size_t)b * C + i] = cs[i];
    if (i >= 1) label2[(size_t)b * (C - 1) + i - 1] = l2[i];
  }
  const int n1 = R - 1, n2 = C - 1;
  float* wb = weights + (size_t)b * n1 * n2;
  for (int e = t; e < n1 * n2; e += 1024) {
    const int c = e % n2 + 1, r = e / n2 + 1;
    float v = m[(size_t)r * C + c];
    const float f1 = l1[r] > 0 ? 1.f : 0.f;
    const float f2 = l2[c] > 0 ? 1.f : 0.f;
    v = (v * f1) * f2;
    wb[e] = sa_pow15(v);
  }
}

#define CAS_LDS_BYTES(R, C) (((size_t)(R) * (C) + 3 * (size_t)(R) + 3 * (size_t)(C)) * 4)
extern "C" int sam6d_coarse_soft_assign(const float* att, int B, int R, int C, float* rmax, float* rsum, float* cmax, float* csum,
                                        int* label1, int* label2, float* weights, float* w1, void* stream) {
  SAM6D_REQUIRE(att && rmax && rsum && cmax && csum && label1 && label2 && weights && w1, "coarse_soft_assign: null pointer");
  SAM6D_REQUIRE(B >= 0 && R >= 2 && C >= 2 && B <= 65535, "coarse_soft_assign: bad sizes");
  SAM6D_REQUIRE(CAS_LDS_BYTES(R, C) <= 160 * 1024,
                "coarse_soft_assign: the %d x %d matrix does not fit the 160 KB of LDS (use sam6d_soft_assign + sam6d_coarse_weights)", R, C);
  if (B == 0) return 0;
  static unsigned long long cas_done = 0;
  if (sam6d_first_use_on_device(&cas_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(coarse_assign_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       160 * 1024);
    if (e != hipSuccess) {
      sam6d_set_error("coarse_soft_assign: cannot reserve LDS: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&cas_done);
  }
  hipLaunchKernelGGL(coarse_assign_kernel, dim3(B), dim3(1024), CAS_LDS_BYTES(R, C), (hipStream_t)stream, att, R, C, rmax, rsum, cmax,
                     csum, label1, label2, weights, w1);
  SAM6D_LAUNCH_CHECK("coarse_soft_assign");
}

// =========================================================================================================
// Weighted sampling: cum = cumsum(w) with a DOUBLE accumulator rounded to float per element (torch CPU cumsum,
// SURVEY 8c n3); cum /= (cum[-1] + 1e-8); idx = first i with cum[i] >= u, 0 if none (model_utils.py:241-243,277-305).
// One workgroup per row: each thread scans a contiguous chunk, chunk totals are scanned in LDS (double).
// =========================================================================================================
// Wave w owns the contiguous segment [w*seg, (w+1)*seg) of the row and walks it 64 elements at a time (coalesced loads): an
// inclusive wave scan in double per step plus the carried sum.  Pass 1 yields the 16 segment totals, pass 2 repeats the same
// arithmetic with the segment's offset as the initial carry, rounds to float, divides by the row total and stores.
// (The previous form gave each thread a contiguous chunk: strided loads and a 1024-wide LDS scan with 20 barriers, 116 us.)
// (round 4: on the DPP path -- Hillis-Steele inside each 16-lane row with zero fill, then the totals of the rows before by row_bcast:15 /
// row_bcast:31 -- instead of six ds_bpermute steps of two registers each; the sums are the same up to the association of the double
// additions, 2^-29 below the float rounding the result goes through)
template <int CTRL, int ROWS, bool ZERO>
__device__ __forceinline__ double dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)u, CTRL, ROWS, 0xf, ZERO);
  const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)(u >> 32), CTRL, ROWS, 0xf, ZERO);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double bcast63_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, 63), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), 63);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_incl_scan_f64(double v, int lane) {
  v += dpp_f64<0x111, 0xf, true>(v);   // row_shr:1
  v += dpp_f64<0x112, 0xf, true>(v);   // row_shr:2
  v += dpp_f64<0x114, 0xf, true>(v);   // row_shr:4
  v += dpp_f64<0x118, 0xf, true>(v);   // row_shr:8
  v += dpp_f64<0x142, 0xa, false>(v);  // row_bcast:15 -> rows 1, 3
  v += dpp_f64<0x143, 0xc, false>(v);  // row_bcast:31 -> rows 2, 3
  return v;
}

__global__ __launch_bounds__(1024) void cumsum_norm_kernel(const float* __restrict__ w, int L, float* __restrict__ cum) {
  __shared__ double s_tot[16];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* x = w + (size_t)b * L;
  float* o = cum + (size_t)b * L;
  const int steps = (L + 1023) / 1024;  // 64-element steps per wave
  const int i0 = wave * steps * 64;
  double carry = 0.0;
  for (int k = 0; k < steps; ++k) {
    const int i = i0 + k * 64 + lane;
    const double inc = wave_incl_scan_f64((i < L) ? (double)x[i] : 0.0, lane);
    carry += bcast63_f64(inc);
  }
  if (lane == 0) s_tot[wave] = carry;
  __syncthreads();
  double off = 0.0, total = 0.0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const double v = s_tot[q];
    off += (q < wave) ? v : 0.0;
    total += v;
  }
  const float den = (float)total + 1e-8f;
  carry = off;
  for (int k = 0; k < steps; ++k) {
    const int i = i0 + k * 64 + lane;
    const double inc = wave_incl_scan_f64((i < L) ? (double)x[i] : 0.0, lane);
    if (i < L) o[i] = (float)(carry + inc) / den;
    carry += bcast63_f64(inc);
  }
}

__global__ __launch_bounds__(256) void sample_first_ge_kernel(const float* __restrict__ cum, const float* __restrict__ u, int L,
                                                              int ns, long total, int* __restrict__ idx) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / ns;
  const float* c = cum + b * L;
  const float v = u[e];
  int lo = 0, hi = L;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (c[mid] >= v) hi = mid; else lo = mid + 1;
  }
  idx[e] = (lo == L) ? 0 : lo;
}

extern "C" int sam6d_weighted_sample(const float* weights, const float* rand, int B, int L, int ns, float* cum_ws, int* idx,
                                     void* stream) {
  SAM6D_REQUIRE(weights && rand && cum_ws && idx && B >= 0 && L > 0 && ns >= 0, "weighted_sample: bad arguments");
  if (B == 0 || ns == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(cumsum_norm_kernel, dim3(B), dim3(1024), 0, s, weights, L, cum_ws);
  const long total = (long)B * ns;
  hipLaunchKernelGGL(sample_first_ge_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, cum_ws, rand, L, ns, total,
                     idx);
  SAM6D_LAUNCH_CHECK("weighted_sample");
}

// =========================================================================================================
// 3x3 rotation from the correlation matrix H = sum src_c^T (w ref_c):  R = V diag(1,1,sign det(V U^T)) U^T.
// Written as R = v1 u1^T + v2 u2^T + (v1 x v2)(u1 x u2)^T over the two dominant singular triplets, which equals the
// reference's formula for any sign convention of the SVD and for rank-2 H (3-point hypotheses always have sigma3 = 0).
// Jacobi eigen-decomposition of H^T H in double.
// =========================================================================================================
__device__ void sym_eig3(double a[3][3], double v[3][3], double w[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 16; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double dia = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-22 * dia) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        const double apq = a[p][q];
        if (fabs(apq) <= 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);
        const double tt = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(tt * tt + 1.0), s = tt * c;
        const int r = 3 - p - q;
        const double app = a[p][p], aqq = a[q][q], arp = a[r][p], arq = a[r][q];
        a[p][p] = app - tt * apq;
        a[q][q] = aqq + tt * apq;
        a[p][q] = a[q][p] = 0.0;
        a[r][p] = a[p][r] = c * arp - s * arq;
        a[r][q] = a[q][r] = s * arp + c * arq;
        for (int i = 0; i < 3; ++i) {
          const double vip = v[i][p], viq = v[i][q];
          v[i][p] = c * vip - s * viq;
          v[i][q] = s * vip + c * viq;
        }
      }
  }
  w[0] = a[0][0];
  w[1] = a[1][1];
  w[2] = a[2][2];
}

__device__ void rotation_from_H(const double H[9], double R[9]) {
  double m[3][3], v[3][3], w[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m[i][j] = H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j] + H[2 * 3 + i] * H[2 * 3 + j];
  sym_eig3(m, v, w);
  int i0 = 0, i1 = 1, i2 = 2;  // sort eigenvalues descending
  if (w[i0] < w[i1]) { int x = i0; i0 = i1; i1 = x; }
  if (w[i0] < w[i2]) { int x = i0; i0 = i2; i2 = x; }
  if (w[i1] < w[i2]) { int x = i1; i1 = i2; i2 = x; }
  double v1[3] = {v[0][i0], v[1][i0], v[2][i0]};
  double v2[3] = {v[0][i1], v[1][i1], v[2][i1]};
  double u1[3], u2[3];
  for (int i = 0; i < 3; ++i) {
    u1[i] = H[i * 3 + 0] * v1[0] + H[i * 3 + 1] * v1[1] + H[i * 3 + 2] * v1[2];
    u2[i] = H[i * 3 + 0] * v2[0] + H[i * 3 + 1] * v2[1] + H[i * 3 + 2] * v2[2];
  }
  const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  if (!(n1 > 1e-150)) {  // H == 0: no information, identity
    for (int i = 0; i < 9; ++i) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    return;
  }
  for (int i = 0; i < 3; ++i) u1[i] /= n1;
  const double d12 = u1[0] * u2[0] + u1[1] * u2[1] + u1[2] * u2[2];
  for (int i = 0; i < 3; ++i) u2[i] -= d12 * u1[i];
  double n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  if (!(n2 > 1e-12 * n1)) {  // rank-1 H (degenerate hypothesis): any unit vector orthogonal to u1 (finite, deterministic)
    const int k = (fabs(u1[0]) <= fabs(u1[1]) && fabs(u1[0]) <= fabs(u1[2])) ? 0 : (fabs(u1[1]) <= fabs(u1[2]) ? 1 : 2);
    double e[3] = {0, 0, 0};
    e[k] = 1.0;
    const double d = u1[k];
    for (int i = 0; i < 3; ++i) u2[i] = e[i] - d * u1[i];
    n2 = sqrt(u2[0] * u2[0] + u2[1] * u2[1] + u2[2] * u2[2]);
  }
  for (int i = 0; i < 3; ++i) u2[i] /= n2;
  const double u3[3] = {u1[1] * u2[2] - u1[2] * u2[1], u1[2] * u2[0] - u1[0] * u2[2], u1[0] * u2[1] - u1[1] * u2[0]};
  const double v3[3] = {v1[1] * v2[2] - v1[2] * v2[1], v1[2] * v2[0] - v1[0] * v2[2], v1[0] * v2[1] - v1[1] * v2[0]};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[i * 3 + j] = v1[i] * u1[j] + v2[i] * u2[j] + v3[i] * u3[j];
}

// The same rotation for a THREE-point hypothesis in closed form (no eigen-iteration).  The centred triples a_k (source) and b_k
// (reference) each span a plane (H = sum w a_k b_k^T has rank <= 2; the 1e-5 in the weight normalisation of model_utils.py:381 leaves a
// third singular value of ~1e-11 that the cross-product form above ignores as well).  With right-handed orthonormal frames E = (e1, e2,
// e1 x e2) of the source plane and F of the reference plane, every candidate is R = F diag(Q, det Q) E^T with Q a 2 x 2 orthogonal
// matrix, and tr(R H) = sum_pq Q_pq M_qp for the 2 x 2 in-plane correlation M_pq = sum_k w (a_k . e_p)(b_k . f_q):
//   det M >= 0:  Q = rotation,   (c, s) ~ (M11 + M22, M12 - M21)      (tr = sigma1 + sigma2)
//   det M <  0:  Q = reflection, (c, s) ~ (M11 - M22, M12 + M21), and the plane normal flips so that R stays proper
// -- the maximiser the SVD formula V diag(1, 1, det(V U^T)) U^T returns.  ~150 fp64 operations and 6 square roots instead of up to 16
// Jacobi sweeps with 6 fp64 divisions / square roots each (coarse_hyp_kernel: 205 -> see DESIGN).  Collinear or coincident triples
// (a sampled pair repeated: rank <= 1, the reference's answer is LAPACK rounding noise) get a frame completed from the coordinate
// axis least aligned with e1 -- finite, deterministic, a proper rotation.
__device__ __forceinline__ void frame3(const double p[3][3], double e[3][3]) {
  double n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) n[k] = p[k][0] * p[k][0] + p[k][1] * p[k][1] + p[k][2] * p[k][2];
  const int k0 = (n[0] >= n[1] && n[0] >= n[2]) ? 0 : (n[1] >= n[2] ? 1 : 2);
  double e1[3] = {1.0, 0.0, 0.0};
  double n1 = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k)
    if (k == k0) {
      n1 = sqrt(n[k]);
      if (n1 > 1e-150) { e1[0] = p[k][0] / n1; e1[1] = p[k][1] / n1; e1[2] = p[k][2] / n1; }
    }
  // the centred vector with the largest part orthogonal to e1
  double q[3] = {0.0, 0.0, 0.0}, qn = -1.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double d = p[k][0] * e1[0] + p[k][1] * e1[1] + p[k][2] * e1[2];
    const double x = p[k][0] - d * e1[0], y = p[k][1] - d * e1[1], z = p[k][2] - d * e1[2];
    const double m = x * x + y * y + z * z;
    if (m > qn) { qn = m; q[0] = x; q[1] = y; q[2] = z; }
  }
  double n2 = sqrt(qn > 0.0 ? qn : 0.0);
  if (!(n2 > 1e-12 * n1) || !(n1 > 1e-150)) {  // collinear / coincident: complete the frame from the axis least aligned with e1
    const int a = (fabs(e1[0]) <= fabs(e1[1]) && fabs(e1[0]) <= fabs(e1[2])) ? 0 : (fabs(e1[1]) <= fabs(e1[2]) ? 1 : 2);
    const double d = e1[a];
    q[0] = (a == 0 ? 1.0 : 0.0) - d * e1[0];
    q[1] = (a == 1 ? 1.0 : 0.0) - d * e1[1];
    q[2] = (a == 2 ? 1.0 : 0.0) - d * e1[2];
    n2 = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
  }
  const double e2[3] = {q[0] / n2, q[1] / n2, q[2] / n2};
#pragma unroll
  for (int i = 0; i < 3; ++i) { e[0][i] = e1[i]; e[1][i] = e2[i]; }
  e[2][0] = e1[1] * e2[2] - e1[2] * e2[1];
  e[2][1] = e1[2] * e2[0] - e1[0] * e2[2];
  e[2][2] = e1[0] * e2[1] - e1[1] * e2[0];
}

// frame whose first axis is the direction of v (completed from the coordinate axis least aligned with it); v = 0 -> the identity frame
__device__ __forceinline__ void frame_from_dir(const double v[3], double e[3][3]) {
  const double n = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
  double e1[3] = {1.0, 0.0, 0.0};
  if (n > 1e-150) { e1[0] = v[0] / n; e1[1] = v[1] / n; e1[2] = v[2] / n; }
  const int a = (fabs(e1[0]) <= fabs(e1[1]) && fabs(e1[0]) <= fabs(e1[2])) ? 0 : (fabs(e1[1]) <= fabs(e1[2]) ? 1 : 2);
  const double d = e1[a];
  double q[3] = {(a == 0 ? 1.0 : 0.0) - d * e1[0], (a == 1 ? 1.0 : 0.0) - d * e1[1], (a == 2 ? 1.0 : 0.0) - d * e1[2]};
  const double n2 = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i) { e[0][i] = e1[i]; e[1][i] = q[i] / n2; }
  e[2][0] = e[0][1] * e[1][2] - e[0][2] * e[1][1];
  e[2][1] = e[0][2] * e[1][0] - e[0][0] * e[1][2];
  e[2][2] = e[0][0] * e[1][1] - e[0][1] * e[1][0];
}

// a, b: the triples centred on their TRUE means ma, mb (see coarse_hyp_kernel)
__device__ void rotation_3pt(const double a[3][3], const double b[3][3], const double ma[3], const double mb[3], double R[9]) {
  double E[3][3], F[3][3];
  double na = 0.0, nb = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    na = fmax(na, a[k][0] * a[k][0] + a[k][1] * a[k][1] + a[k][2] * a[k][2]);
    nb = fmax(nb, b[k][0] * b[k][0] + b[k][1] * b[k][1] + b[k][2] * b[k][2]);
  }
  if (!(na > 1e-300) || !(nb > 1e-300)) {
    // one of the triples is a single point sampled three times: H0 = 0, and what is left of the reference's H is the rank-1 trace of its
    // shrunk centroids, 3 w d_a d_b^T with d_a ~ ma, d_b ~ mb -- its rotation turns the direction of ma into the direction of mb (twist
    // undetermined).  The residual the caller computes depends on exactly that alignment, so it is reproduced.
    frame_from_dir(ma, E);
    frame_from_dir(mb, F);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) R[i * 3 + j] = F[0][i] * E[0][j] + F[1][i] * E[1][j] + F[2][i] * E[2][j];
    return;
  }
  frame3(a, E);
  frame3(b, F);
  double M[2][2] = {{0.0, 0.0}, {0.0, 0.0}};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double a1 = a[k][0] * E[0][0] + a[k][1] * E[0][1] + a[k][2] * E[0][2];
    const double a2 = a[k][0] * E[1][0] + a[k][1] * E[1][1] + a[k][2] * E[1][2];
    const double b1 = b[k][0] * F[0][0] + b[k][1] * F[0][1] + b[k][2] * F[0][2];
    const double b2 = b[k][0] * F[1][0] + b[k][1] * F[1][1] + b[k][2] * F[1][2];
    M[0][0] += a1 * b1; M[0][1] += a1 * b2; M[1][0] += a2 * b1; M[1][1] += a2 * b2;
  }
  const bool rot = (M[0][0] * M[1][1] - M[0][1] * M[1][0]) >= 0.0;
  double c = rot ? (M[0][0] + M[1][1]) : (M[0][0] - M[1][1]);
  double s = rot ? (M[0][1] - M[1][0]) : (M[0][1] + M[1][0]);
  const double r = sqrt(c * c + s * s);
  if (r > 1e-300) { c /= r; s /= r; } else { c = 1.0; s = 0.0; }
  // G = [[c, -s, 0], [s, c, 0], [0, 0, 1]] (rotation) or [[c, s, 0], [s, -c, 0], [0, 0, -1]] (reflection in the plane, normal flipped)
  const double g00 = c, g01 = rot ? -s : s, g10 = s, g11 = rot ? c : -c, g22 = rot ? 1.0 : -1.0;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      R[i * 3 + j] = F[0][i] * (g00 * E[0][j] + g01 * E[1][j]) + F[1][i] * (g10 * E[0][j] + g11 * E[1][j]) + F[2][i] * g22 * E[2][j];
}

// Coarse hypotheses (model_utils.py:244-257): sample s = 3h+k picks the pair (i1 = idx / N2, i2 = idx % N2);
// R,t = procrustes(src = p2 triple -> ref = p1 triple), unit weights (thresh 0.5 keeps them), eps 1e-5;
// dis = mean_k |(p1_k - t) R - p2_k|.
__global__ __launch_bounds__(256) void coarse_hyp_kernel(const int* __restrict__ idx, const float* __restrict__ pts1,
                                                         const float* __restrict__ pts2, int N1, int N2, int nh, long total,
                                                         float* __restrict__ Rs, float* __restrict__ ts, float* __restrict__ dis) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / nh;
  const float* P1 = pts1 + b * N1 * 3;
  const float* P2 = pts2 + b * N2 * 3;
  double p1[3][3], p2[3][3];
  for (int k = 0; k < 3; ++k) {
    const int id = idx[e * 3 + k];
    const int a = min(id / N2, N1 - 1), c = min(id % N2, N2 - 1);
    for (int d = 0; d < 3; ++d) {
      p1[k][d] = (double)P1[a * 3 + d];
      p2[k][d] = (double)P2[c * 3 + d];
    }
  }
  const double w = (double)(1.0f / (3.0f + 1e-5f));
  double sc[3], rc[3];
  for (int d = 0; d < 3; ++d) {
    sc[d] = (p2[0][d] + p2[1][d] + p2[2][d]) * w;
    rc[d] = (p1[0][d] + p1[1][d] + p1[2][d]) * w;
  }
  // H = sum_k (p2_k - sc) (w (p1_k - rc))^T (model_utils.py:384-389).  sc, rc are the reference's centroids -- sums times w = 1 / (3 + 1e-5),
  // i.e. the true means shrunk by 3.3e-6 -- so its centred vectors carry a common offset d_a, d_b (up to 2.7e-5 for a cloud at z = 8).  In H
  // the offsets cancel to first order (H = w H0 + 3 w d_a d_b^T with H0 from the TRUE means: the cross terms multiply sum_k (p_k - mean) = 0),
  // so the rotation of H is the rotation of H0 to ~1e-10.  The closed form builds plane frames from the vectors themselves, where the
  // offset would tilt the plane by d / |a| ~ 1e-4: it takes the vectors centred on the true means.
  double ca[3][3], cb[3][3], ma[3], mb[3];
  for (int d = 0; d < 3; ++d) {
    ma[d] = (p2[0][d] + p2[1][d] + p2[2][d]) / 3.0;
    mb[d] = (p1[0][d] + p1[1][d] + p1[2][d]) / 3.0;
    for (int k = 0; k < 3; ++k) {
      ca[k][d] = p2[k][d] - ma[d];
      cb[k][d] = p1[k][d] - mb[d];
    }
  }
  double R[9], t[3];
  rotation_3pt(ca, cb, ma, mb, R);
  for (int i = 0; i < 3; ++i) t[i] = rc[i] - (R[i * 3] * sc[0] + R[i * 3 + 1] * sc[1] + R[i * 3 + 2] * sc[2]);
  double acc = 0.0;
  for (int k = 0; k < 3; ++k) {
    double r2 = 0.0;
    for (int j = 0; j < 3; ++j) {
      const double y = (p1[k][0] - t[0]) * R[0 * 3 + j] + (p1[k][1] - t[1]) * R[1 * 3 + j] + (p1[k][2] - t[2]) * R[2 * 3 + j];
      const double dd = y - p2[k][j];
      r2 += dd * dd;
    }
    acc += sqrt(r2);
  }
  for (int i = 0; i < 9; ++i) Rs[e * 9 + i] = (float)R[i];
  for (int i = 0; i < 3; ++i) ts[e * 3 + i] = (float)t[i];
  dis[e] = (float)(acc / 3.0);
}

extern "C" int sam6d_coarse_hypotheses(const int* idx, const float* pts1, const float* pts2, int B, int N1, int N2, int nh,
                                       float* Rs, float* ts, float* dis, void* stream) {
  SAM6D_REQUIRE(idx && pts1 && pts2 && Rs && ts && dis && B >= 0 && N1 > 0 && N2 > 0 && nh >= 0, "coarse_hypotheses: bad arguments");
  const long total = (long)B * nh;
  if (total == 0) return 0;
  hipLaunchKernelGGL(coarse_hyp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, idx, pts1,
                     pts2, N1, N2, nh, total, Rs, ts, dis);
  SAM6D_LAUNCH_CHECK("coarse_hypotheses");
}

// k smallest of each row by rank counting: rank_i = #{j : d_j < d_i or (d_j == d_i and j < i)}; sel[rank] = i.
// Output is sorted ascending (torch.topk(largest=False) order; tie order there is unspecified).
// A wave holds 64 consecutive candidates i0 .. i0+63: every j below i0 precedes all of them (count d_j <= d_i), every j from
// i0+64 on follows all of them (count d_j < d_i); only the 64 j inside the wave's own range need the full tie rule.  Two VALU
// instructions per (i, j) pair instead of six, four j per broadcast LDS read.
__global__ __launch_bounds__(256) void select_smallest_kernel(const float* __restrict__ dis, int n, int k, int* __restrict__ sel) {
  extern __shared__ __attribute__((aligned(16))) float sd[];  // n floats, padded with +inf to a multiple of 4
  const int b = blockIdx.y;
  const float* d = dis + (size_t)b * n;
  const int n4 = (n + 3) & ~3;
  for (int i = threadIdx.x; i < n4; i += 256) sd[i] = (i < n) ? d[i] : INFINITY;
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int i0 = i & ~63;  // first candidate of this wave (multiple of 64, so of 4)
  if (i0 >= n) return;
  const float v = (i < n) ? sd[i] : INFINITY;
  // sign bit of (v - o) = [o > v], of (o - v) = [o < v]: one v_sub_f32 and one v_alignbit_b32 (shift the bit into a 32-bit
  // register) per pair, one popcount per 32 pairs -- compare + conditional add costs three VALU slots plus wait states
  auto signs4 = [](unsigned int acc, float a, const float4& q, bool q_minus_a) {
    const float d0 = q_minus_a ? q.x - a : a - q.x, d1 = q_minus_a ? q.y - a : a - q.y;
    const float d2 = q_minus_a ? q.z - a : a - q.z, d3 = q_minus_a ? q.w - a : a - q.w;
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d0), 31);
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d1), 31);
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d2), 31);
    acc = __builtin_amdgcn_alignbit(acc, __float_as_uint(d3), 31);
    return acc;
  };
  int gt_before = 0;  // #{j < i0 : d_j > v}; the j below the wave's range that count are the other i0 - gt_before
  for (int j = 0; j < i0; j += 32) {
    unsigned int acc = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = signs4(acc, v, *reinterpret_cast<const float4*>(&sd[j + 4 * u]), false);
    gt_before += __popc(acc);
  }
  int rank = i0 - gt_before;
  const int i1 = min(i0 + 64, n4);
  for (int j = i0; j < i1; ++j) {
    const float o = sd[j];
    rank += (o < v || (o == v && j < i)) ? 1 : 0;
  }
  int j = i1;
  for (; j + 32 <= n4; j += 32) {
    unsigned int acc = 0u;
#pragma unroll
    for (int u = 0; u < 8; ++u) acc = signs4(acc, v, *reinterpret_cast<const float4*>(&sd[j + 4 * u]), true);
    rank += __popc(acc);
  }
  for (; j < n4; j += 4) {  // the +inf padding never counts: inf - v is +inf
    const float4 o = *reinterpret_cast<const float4*>(&sd[j]);
    rank += (o.x < v) + (o.y < v) + (o.z < v) + (o.w < v);
  }
  if (i < n && rank < k) sel[(size_t)b * k + rank] = i;
}

// The same selection in O(n) instead of O(n^2): a three-level radix select (11 + 11 + 10 bits of the order-preserving integer image of
// the floats) finds the k-th smallest value T and the number of smaller elements, the elements below T plus the first (k - that
// number) elements EQUAL to T in index order are compacted, and only those k candidates are ranked against each other (the tie rule of
// the rank counting above: equal values in index order).  Identical output, one workgroup per row: 6000 -> 300 in ~15 us instead of
// 72 us of all-pairs comparisons.
#define SR_T 1024
__device__ __forceinline__ unsigned sr_key(float d) {
  if (d == 0.0f) d = 0.0f;  // -0 and +0 compare equal in the rank rule
  const unsigned u = __float_as_uint(d);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
#define SR_STATIC_LDS (2048 * 4 + 64)  // hist[] + the three scalars of select_radix_kernel, padded
__global__ __launch_bounds__(SR_T) void select_radix_kernel(const float* __restrict__ dis, int n, int k, int* __restrict__ sel) {
  extern __shared__ unsigned sr_keys[];        // n keys, then k candidate ids, then k candidate keys
  __shared__ int hist[2048];
  __shared__ int s_bin, s_before, s_cnt;
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
  unsigned* cand_id = sr_keys + n;
  unsigned* cand_key = cand_id + k;
  const float* d = dis + (size_t)b * n;
  for (int i = t; i < n; i += SR_T) sr_keys[i] = sr_key(d[i]);
  // level L: histogram of the bits [shift, shift + bits) of the keys whose higher bits equal `prefix`; the bin in which the cumulative
  // count (plus `before`, the elements already known to be smaller) first exceeds k - 1 holds the k-th smallest
  unsigned prefix = 0;
  int before = 0;
  const int shifts[3] = {21, 10, 0}, nbits[3] = {11, 11, 10};
  for (int L = 0; L < 3; ++L) {
    const int shift = shifts[L], bins = 1 << nbits[L];
    for (int i = t; i < 2048; i += SR_T) hist[i] = 0;
    __syncthreads();
    for (int i = t; i < n; i += SR_T) {
      const unsigned u = sr_keys[i];
      const bool in = (L == 0) || ((u >> (shift + nbits[L])) == prefix);
      if (in) atomicAdd(&hist[(u >> shift) & (bins - 1)], 1);
    }
    __syncthreads();
    if (t < 64) {  // one wave scans the bins: 32 per lane, then a wave prefix sum
      const int per = bins / 64;
      int sum = 0;
      for (int j = 0; j < per; ++j) sum += hist[t * per + j];
      int inc = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int nb = __shfl_up(inc, o, 64);
        if (lane >= o) inc += nb;
      }
      const int excl = inc - sum;
      const int need = k - 1 - before;  // 0-based rank of the wanted element among the elements of this level
      if (need >= excl && need < inc) {  // exactly one lane
        int run = excl;
        for (int j = 0; j < per; ++j) {
          const int c = hist[t * per + j];
          if (need < run + c) { s_bin = t * per + j; s_before = before + run; break; }
          run += c;
        }
      }
    }
    __syncthreads();
    prefix = (prefix << nbits[L]) | (unsigned)s_bin;
    before = s_before;
    __syncthreads();
  }
  const unsigned T = prefix;            // the k-th smallest key; `before` keys are smaller
  const int need_eq = k - before;       // >= 1 elements equal to T are taken, lowest indices first
  if (t == 0) s_cnt = 0;
  __syncthreads();
  // keys below T: compacted in any order
  for (int i = t; i < n; i += SR_T) {
    const unsigned u = sr_keys[i];
    if (u < T) {
      const int slot = atomicAdd(&s_cnt, 1);
      cand_id[slot] = (unsigned)i;
      cand_key[slot] = u;
    }
  }
  // keys equal to T: the first need_eq in index order (one wave walks the row, ballot + prefix popcount keeps the order)
  if (t < 64) {
    int taken = 0;
    for (int i0 = 0; i0 < n && taken < need_eq; i0 += 64) {
      const int i = i0 + lane;
      const bool eq = i < n && sr_keys[i] == T;
      const unsigned long long m = __ballot(eq);
      const int pos = taken + __popcll(m & ((1ull << lane) - 1ull));
      if (eq && pos < need_eq) {
        cand_id[before + pos] = (unsigned)i;
        cand_key[before + pos] = T;
      }
      taken += __popcll(m);
    }
  }
  __syncthreads();
  // rank the k candidates among themselves (value, then index)
  for (int c = t; c < k; c += SR_T) {
    const unsigned u = cand_key[c], id = cand_id[c];
    int rank = 0;
    for (int j = 0; j < k; ++j) {
      const unsigned uj = cand_key[j];
      rank += (uj < u || (uj == u && cand_id[j] < id)) ? 1 : 0;
    }
    sel[(size_t)b * k + rank] = (int)id;
  }
}

extern "C" int sam6d_select_smallest(const float* dis, int B, int n, int k, int* sel, void* stream) {
  SAM6D_REQUIRE(dis && sel && B >= 0 && n > 0 && k > 0 && k <= n && n <= 15000 && B <= 65535, "select_smallest: bad arguments (n <= 15000)");
  if (B == 0) return 0;
  static int use_radix = -1;
  if (use_radix < 0) {
    const char* e = getenv("SAM6D_SELECT_RADIX");  // A/B switch: 0 = the all-pairs rank counting
    use_radix = (e && e[0] == '0') ? 0 : 1;
  }
  // dynamic + static LDS (hist[2048] + 3 words, rounded up) within the default 64 KB limit; longer rows keep the all-pairs kernel
  if (use_radix && (size_t)(n + 2 * k) * 4 + SR_STATIC_LDS <= 65536)
    hipLaunchKernelGGL(select_radix_kernel, dim3(B), dim3(SR_T), (size_t)(n + 2 * k) * 4, (hipStream_t)stream, dis, n, k, sel);
  else
    hipLaunchKernelGGL(select_smallest_kernel, dim3(cdiv(n, 256), B), dim3(256), (size_t)((n + 3) & ~3) * 4, (hipStream_t)stream, dis, n, k,
                       sel);
  SAM6D_LAUNCH_CHECK("select_smallest");
}

// Hypothesis scoring (model_utils.py:261-267): score = sum(w1) / (sum_i w1_i * min_m |(p1_i - t) R - model_m| + 1e-8),
// distances in the pairwise_distance bit recipe, model = model_raw / (radius + 1e-6) (coarse_point_matching.py:60).
// One workgroup scores SH_G hypotheses: SH_G * N1 (hypothesis, point) items are dealt to the 256 threads (98 % lane
// utilisation at N1 = 196 instead of 77 % with one hypothesis per workgroup) and the LDS copy of the CAD points is
// shared by the group.  Reductions are in a fixed order (wave butterfly, then waves 0..3): scores are reproducible.
#define SH_G 4
__global__ __launch_bounds__(256) void score_hyp_kernel(const int* __restrict__ sel, const float* __restrict__ Rs,
                                                        const float* __restrict__ ts, const float* __restrict__ pts1,
                                                        const float* __restrict__ w1, const float* __restrict__ model,
                                                        const float* __restrict__ radius, int N1, int P, int nh, int k,
                                                        float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [P*4]: x,y,z,|m|^2
  __shared__ float sRt[SH_G][12];
  __shared__ float red[2][SH_G][4];
  const int b = blockIdx.y, s0 = blockIdx.x * SH_G, t = threadIdx.x;
  const float den = radius[b] + 1e-6f;
  const float* mb = model + (size_t)b * P * 3;
  for (int i = t; i < P; i += 256) {
    const float x = mb[i * 3] / den, y = mb[i * 3 + 1] / den, z = mb[i * 3 + 2] / den;
    sm[i * 4] = x; sm[i * 4 + 1] = y; sm[i * 4 + 2] = z; sm[i * 4 + 3] = sqnorm3(x, y, z);
  }
  if (t < SH_G * 12) {
    const int g = t / 12, e = t % 12;
    const int s = min(s0 + g, k - 1);
    const int h = sel[(size_t)b * k + s];
    sRt[g][e] = (e < 9) ? Rs[((size_t)b * nh + h) * 9 + e] : ts[((size_t)b * nh + h) * 3 + (e - 9)];
  }
  __syncthreads();
  float sw[SH_G], sdw[SH_G];
#pragma unroll
  for (int g = 0; g < SH_G; ++g) sw[g] = sdw[g] = 0.f;
  for (int item = t; item < SH_G * N1; item += 256) {
    const int g = item / N1, i = item - g * N1;
    const float* R = sRt[g];
    const float* p = pts1 + ((size_t)b * N1 + i) * 3;
    const float d0 = p[0] - R[9], d1 = p[1] - R[10], d2 = p[2] - R[11];
    const float x0 = fmaf(d2, R[6], fmaf(d1, R[3], d0 * R[0]));
    const float x1 = fmaf(d2, R[7], fmaf(d1, R[4], d0 * R[1]));
    const float x2 = fmaf(d2, R[8], fmaf(d1, R[5], d0 * R[2]));
    const float sx = sqnorm3(x0, x1, x2);
    // min over the CAD points of pdist3, six VALU instructions per pair instead of eight, same bits: 2*xy is exact, so
    // sx - 2*xy = fma(-2, xy, sx); and min_m max(d_m, 0) = max(min_m d_m, 0), so the clamp moves out of the loop
    auto raw = [&](float qx, float qy, float qz, float qw) {
      const float xy = fmaf(x2, qz, fmaf(x1, qy, x0 * qx));
      return fmaf(-2.0f, xy, sx) + qw;
    };
    float mn0 = INFINITY, mn1 = INFINITY, mn2 = INFINITY, mn3 = INFINITY;
    int m = 0;
    for (; m + 4 <= P; m += 4) {
      const float4 q0 = *reinterpret_cast<const float4*>(&sm[m * 4]);
      const float4 q1 = *reinterpret_cast<const float4*>(&sm[m * 4 + 4]);
      const float4 q2 = *reinterpret_cast<const float4*>(&sm[m * 4 + 8]);
      const float4 q3 = *reinterpret_cast<const float4*>(&sm[m * 4 + 12]);
      mn0 = fminf(mn0, raw(q0.x, q0.y, q0.z, q0.w));
      mn1 = fminf(mn1, raw(q1.x, q1.y, q1.z, q1.w));
      mn2 = fminf(mn2, raw(q2.x, q2.y, q2.z, q2.w));
      mn3 = fminf(mn3, raw(q3.x, q3.y, q3.z, q3.w));
    }
    for (; m < P; ++m) {
      const float4 q = *reinterpret_cast<const float4*>(&sm[m * 4]);
      mn0 = fminf(mn0, raw(q.x, q.y, q.z, q.w));
    }
    float mn = fminf(fminf(mn0, mn1), fminf(mn2, mn3));
    mn = mn < 0.0f ? 0.0f : mn;
    const float wv = w1[(size_t)b * N1 + i];
#pragma unroll
    for (int gg = 0; gg < SH_G; ++gg)
      if (gg == g) {
        sw[gg] += wv;
        sdw[gg] += sqrtf(mn) * wv;
      }
  }
#pragma unroll
  for (int g = 0; g < SH_G; ++g) {
    const float a = wave_sum(sw[g]), c = wave_sum(sdw[g]);
    if ((t & 63) == 0) {
      red[0][g][t >> 6] = a;
      red[1][g][t >> 6] = c;
    }
  }
  __syncthreads();
  if (t < SH_G && s0 + t < k) {
    const float a = (red[0][t][0] + red[0][t][1]) + (red[0][t][2] + red[0][t][3]);
    const float c = (red[1][t][0] + red[1][t][1]) + (red[1][t][2] + red[1][t][3]);
    scores[(size_t)b * k + s0 + t] = a / (c + 1e-8f);
  }
}

// argmax over the k scored hypotheses (first maximum, model_utils.py:268) -> R (B,3,3), t (B,3)
__global__ __launch_bounds__(64) void pick_best_kernel(const float* __restrict__ scores, const int* __restrict__ sel,
                                                       const float* __restrict__ Rs, const float* __restrict__ ts, int nh, int k,
                                                       float* __restrict__ R, float* __restrict__ t, int* __restrict__ best_out) {
  const int b = blockIdx.x, lane = threadIdx.x;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int i = lane; i < k; i += 64) {
    const float v = scores[(size_t)b * k + i];
    if (v > best) { best = v; bi = i; }
  }
  wave_argmax_first(best, bi);
  if (bi == 0x7fffffff) bi = 0;
  const int h = sel[(size_t)b * k + bi];
  if (lane < 9) R[b * 9 + lane] = Rs[((size_t)b * nh + h) * 9 + lane];
  if (lane < 3) t[b * 3 + lane] = ts[((size_t)b * nh + h) * 3 + lane];
  if (lane == 0 && best_out) best_out[b] = h;
}

// ---- the same scoring on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), bit for bit the recipe above.
// The K = 3 contraction x . y of pairwise_distance (model_utils.py:117: torch.matmul) is what the reference itself runs as a GEMM.  One
// 32x32x2 instruction is the k-ordered chain fma(a1, b1, fma(a0, b0, c)) (scratch/ubench/mfma_f32_chain.hip: 0 mismatches), so with
//     A (CAD point m)   = (-2 y0, -2 y1 | -2 y2, 1)        B (item = hypothesis, scene point) = (x0, x1 | x2, |x|^2)
// two instructions on a zero accumulator give  fma(1, |x|^2, fma(-2 y2, x2, fma(-2 y1, x1, rn(-2 y0 x0))))  =  rn(|x|^2 - 2 xy)  with
// xy = fma(x2, y2, fma(x1, y1, rn(x0 y0))) (scaling by -2 commutes with every rounding): exactly `fmaf(-2, xy, sx)` of the VALU kernel
// for 1024 pairs per two instructions.  What is left for the vector ALU per pair: + |y|^2 and the running minimum (1.5 instructions
// instead of 6).  A wave owns SM_CT column tiles of 32 items and walks the 32-row tiles of the CAD points, whose A operands and norms it
// loads once per row tile for all its column tiles.  Items are (rank s of the hypothesis, scene point i) in one linear order per
// proposal, so no lane idles at N1 = 196; the weighted distances go to a workspace and are summed per hypothesis in a fixed order by
// score_sum_kernel; pick_best_kernel takes the arg-max (first maximum, model_utils.py:268).
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define SM_CT 8
#define SM_WAVES 4
__global__ __launch_bounds__(SM_WAVES * 64) void score_hyp_mfma_kernel(const int* __restrict__ sel, const float* __restrict__ Rs,
                                                                      const float* __restrict__ ts, const float* __restrict__ pts1,
                                                                      const float* __restrict__ w1, const float* __restrict__ model,
                                                                      const float* __restrict__ radius, int N1, int P, int Ppad, int nh,
                                                                      int k, float* __restrict__ dw) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][Ppad]: -2 y0 | -2 y1 | -2 y2 | 1;  then [Ppad]: |y|^2 (+inf padding)
  const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6, j = lane & 31, kk = lane >> 5;
  const float den = radius[b] + 1e-6f;
  const float* mb = model + (size_t)b * P * 3;
  for (int i = t; i < Ppad; i += SM_WAVES * 64) {
    float x = 0.f, y = 0.f, z = 0.f, sq = INFINITY;
    if (i < P) {
      x = mb[i * 3] / den; y = mb[i * 3 + 1] / den; z = mb[i * 3 + 2] / den;
      sq = sqnorm3(x, y, z);
    }
    sm[i] = -2.0f * x; sm[Ppad + i] = -2.0f * y; sm[2 * Ppad + i] = -2.0f * z; sm[3 * Ppad + i] = 1.0f;
    sm[4 * Ppad + i] = sq;
  }
  const long total = (long)k * N1;
  const long e0 = ((long)(blockIdx.x * SM_WAVES + wave) * SM_CT) * 32 + j;
  float b1[SM_CT], b2[SM_CT], mn[SM_CT];
#pragma unroll
  for (int c = 0; c < SM_CT; ++c) {
    const long e = e0 + 32 * c;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, sx = 0.f;
    if (e < total) {
      const int s = (int)(e / N1), i = (int)(e - (long)s * N1);
      const int h = sel[(size_t)b * k + s];
      const float* R = Rs + ((size_t)b * nh + h) * 9;
      const float* T = ts + ((size_t)b * nh + h) * 3;
      const float* p = pts1 + ((size_t)b * N1 + i) * 3;
      const float d0 = p[0] - T[0], d1 = p[1] - T[1], d2 = p[2] - T[2];
      x0 = fmaf(d2, R[6], fmaf(d1, R[3], d0 * R[0]));
      x1 = fmaf(d2, R[7], fmaf(d1, R[4], d0 * R[1]));
      x2 = fmaf(d2, R[8], fmaf(d1, R[5], d0 * R[2]));
      sx = sqnorm3(x0, x1, x2);
    }
    b1[c] = kk ? x1 : x0;
    b2[c] = kk ? sx : x2;
    mn[c] = INFINITY;
  }
  __syncthreads();
  f32x16 zero;
#pragma unroll
  for (int v = 0; v < 16; ++v) zero[v] = 0.f;
  const float* a1p = sm + kk * Ppad + j;
  const float* a2p = sm + (2 + kk) * Ppad + j;
  const float* syp = sm + 4 * Ppad + kk * 4;
  for (int r = 0; r < Ppad; r += 32) {
    const float a1 = a1p[r], a2 = a2p[r];
    const float4 s0 = *reinterpret_cast<const float4*>(syp + r), s1 = *reinterpret_cast<const float4*>(syp + r + 8);
    const float4 s2 = *reinterpret_cast<const float4*>(syp + r + 16), s3 = *reinterpret_cast<const float4*>(syp + r + 24);
    // software pipeline over the column tiles: the two instructions of tile c + 1 are issued before the vector work on tile c, into
    // the other of two accumulator sets (one set: MFMA -> MFMA -> wait -> 26 vector instructions, strictly in series per wave)
    f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[0], zero, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2[0], acc, 0, 0, 0);
#pragma unroll
    for (int c = 0; c < SM_CT; ++c) {
      f32x16 nxt = zero;
      if (c + 1 < SM_CT) {
        nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[c + 1], zero, 0, 0, 0);
        nxt = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2[c + 1], nxt, 0, 0, 0);
      }
      // accumulator register v of this lane: CAD row r + 8 (v >> 2) + 4 kk + (v & 3), column j
      const float m0 = fminf(fminf(acc[0] + s0.x, acc[1] + s0.y), fminf(acc[2] + s0.z, acc[3] + s0.w));
      const float m1 = fminf(fminf(acc[4] + s1.x, acc[5] + s1.y), fminf(acc[6] + s1.z, acc[7] + s1.w));
      const float m2 = fminf(fminf(acc[8] + s2.x, acc[9] + s2.y), fminf(acc[10] + s2.z, acc[11] + s2.w));
      const float m3 = fminf(fminf(acc[12] + s3.x, acc[13] + s3.y), fminf(acc[14] + s3.z, acc[15] + s3.w));
      mn[c] = fminf(fminf(mn[c], m0), fminf(fminf(m1, m2), m3));
      acc = nxt;
    }
  }
#pragma unroll
  for (int c = 0; c < SM_CT; ++c) {
    float m = fminf(mn[c], __shfl_xor(mn[c], 32, 64));
    m = m < 0.0f ? 0.0f : m;
    const long e = e0 + 32 * c;
    if (kk == 0 && e < total) {
      const int i = (int)(e % N1);
      dw[(size_t)b * total + e] = sqrtf(m) * w1[(size_t)b * N1 + i];
    }
  }
}

// scores[s] = sum_i w1_i / (sum_i dw[s][i] + 1e-8) (model_utils.py:266-267), a wave per hypothesis: lane-strided partial sums, then the
// wave butterfly -- a fixed order.  pick_best_kernel then takes the first maximum and its pose (model_utils.py:268-275).
__global__ __launch_bounds__(256) void score_sum_kernel(const float* __restrict__ dw, const float* __restrict__ w1, int N1, int k,
                                                        float* __restrict__ scores) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, s = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (s >= k) return;
  const float* d = dw + ((size_t)b * k + s) * N1;
  float a = 0.f, c = 0.f;
  for (int i = lane; i < N1; i += 64) {
    a += w1[(size_t)b * N1 + i];
    c += d[i];
  }
  a = wave_sum(a);
  c = wave_sum(c);
  if (lane == 0) scores[(size_t)b * k + s] = a / (c + 1e-8f);
}

extern "C" size_t sam6d_score_select_workspace_bytes(int B, int N1, int k) { return (size_t)B * N1 * k * 4; }

extern "C" int sam6d_score_select_hypotheses_ws(const int* sel, const float* Rs, const float* ts, const float* pts1, const float* w1,
                                                const float* model, const float* radius, int B, int N1, int P, int nh, int k,
                                                float* scores, float* R, float* t, int* best, float* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(sel && Rs && ts && pts1 && w1 && model && radius && scores && R && t && ws, "score_select_hypotheses_ws: null pointer");
  SAM6D_REQUIRE(B >= 0 && N1 > 0 && P > 0 && P <= 4096 && k > 0 && B <= 65535, "score_select_hypotheses_ws: bad sizes (P <= 4096)");
  SAM6D_REQUIRE(ws_bytes >= (size_t)B * N1 * k * 4, "score_select_hypotheses_ws: workspace too small (sam6d_score_select_workspace_bytes)");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int Ppad = (P + 31) & ~31;
  const long tiles = ((long)k * N1 + 31) / 32;
  const int wgs = (int)((tiles + SM_WAVES * SM_CT - 1) / (SM_WAVES * SM_CT));
  static unsigned long long shm_done = 0;  // P = 4096 needs 80 KB of dynamic LDS: above the default 64 KB limit
  if (sam6d_first_use_on_device(&shm_done)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(score_hyp_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       4096 * 20);
    if (e != hipSuccess) {
      sam6d_set_error("score_select_hypotheses_ws: cannot reserve LDS: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&shm_done);
  }
  hipLaunchKernelGGL(score_hyp_mfma_kernel, dim3(wgs, B), dim3(SM_WAVES * 64), (size_t)Ppad * 20, s, sel, Rs, ts, pts1, w1, model, radius, N1,
                     P, Ppad, nh, k, ws);
  SAM6D_LAUNCH_CHECK_CONT("score_select_hypotheses_ws(score)");
  hipLaunchKernelGGL(score_sum_kernel, dim3(cdiv(k, 4), B), dim3(256), 0, s, ws, w1, N1, k, scores);
  SAM6D_LAUNCH_CHECK_CONT("score_select_hypotheses_ws(sum)");
  hipLaunchKernelGGL(pick_best_kernel, dim3(B), dim3(64), 0, s, scores, sel, Rs, ts, nh, k, R, t, best);
  SAM6D_LAUNCH_CHECK("score_select_hypotheses_ws");
}

extern "C" int sam6d_score_select_hypotheses(const int* sel, const float* Rs, const float* ts, const float* pts1, const float* w1,
                                             const float* model, const float* radius, int B, int N1, int P, int nh, int k,
                                             float* scores, float* R, float* t, int* best, void* stream) {
  SAM6D_REQUIRE(sel && Rs && ts && pts1 && w1 && model && radius && scores && R && t, "score_select_hypotheses: null pointer");
  SAM6D_REQUIRE(B >= 0 && N1 > 0 && P > 0 && P <= 8192 && k > 0 && B <= 65535, "score_select_hypotheses: bad sizes (P <= 8192)");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(score_hyp_kernel, dim3(cdiv(k, SH_G), B), dim3(256), (size_t)P * 16, s, sel, Rs, ts, pts1, w1, model, radius, N1, P, nh, k,
                     scores);
  hipLaunchKernelGGL(pick_best_kernel, dim3(B), dim3(64), 0, s, scores, sel, Rs, ts, nh, k, R, t, best);
  SAM6D_LAUNCH_CHECK("score_select_hypotheses");
}

// =========================================================================================================
// Fine pose (model_utils.py:308-341)
// =========================================================================================================
// per row r >= 1:  A[r,c] = S[r,c] * [l1>0] * [l2>0]  (c >= 1);  weight = sum_c A;  pred = sum_c (A / (weight + 1e-6)) p2[c-1]
template <bool FAST>
__global__ __launch_bounds__(256) void fine_assign_kernel(const float* __restrict__ att, int R, int C, long rows,
                                                          const float* __restrict__ rmax, const float* __restrict__ rsum,
                                                          const float* __restrict__ cmax, const float* __restrict__ csum,
                                                          const int* __restrict__ label1, const int* __restrict__ label2,
                                                          const float* __restrict__ pts2, float* __restrict__ pred,
                                                          float* __restrict__ weight) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over B*(R-1)
  if (w >= rows) return;
  const int lane = threadIdx.x & 63;
  const long b = w / (R - 1);
  const int r = (int)(w % (R - 1)) + 1;
  const float* a = att + ((size_t)b * R + r) * C;
  const float rm = rmax[b * R + r], rs = rsum[b * R + r];
  const float* cm = cmax + b * C;
  const float* cs = csum + b * C;
  const int* l2 = label2 + b * (C - 1);
  const float* p2 = pts2 + b * (C - 1) * 3;
  const float f1 = label1[w] > 0 ? 1.f : 0.f;
  float sa = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
  for (int c = 1 + lane; c < C; c += 64) {
    float v = sa_val<FAST>(a[c], rm, rs, cm[c], cs[c]);
    v = (v * f1) * (l2[c - 1] > 0 ? 1.f : 0.f);
    sa += v;
    sx = fmaf(v, p2[(c - 1) * 3], sx);
    sy = fmaf(v, p2[(c - 1) * 3 + 1], sy);
    sz = fmaf(v, p2[(c - 1) * 3 + 2], sz);
  }
  sa = wave_sum(sa); sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
  if (lane == 0) {
    const float den = sa + 1e-6f;
    weight[w] = sa;
    pred[w * 3] = sx / den;
    pred[w * 3 + 1] = sy / den;
    pred[w * 3 + 2] = sz / den;
  }
}

extern "C" int sam6d_fine_assign(const float* att, int B, int R, int C, const float* rmax, const float* rsum, const float* cmax,
                                 const float* csum, const int* label1, const int* label2, const float* pts2, float* pred,
                                 float* weight, void* stream) {
  SAM6D_REQUIRE(att && rmax && rsum && cmax && csum && label1 && label2 && pts2 && pred && weight, "fine_assign: null pointer");
  const long rows = (long)B * (R - 1);
  if (rows == 0) return 0;
  if (R > 256)
    hipLaunchKernelGGL(fine_assign_kernel<true>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, att, R, C, rows,
                       rmax, rsum, cmax, csum, label1, label2, pts2, pred, weight);
  else
    hipLaunchKernelGGL(fine_assign_kernel<false>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, att, R, C, rows,
                       rmax, rsum, cmax, csum, label1, label2, pts2, pred, weight);
  SAM6D_LAUNCH_CHECK("fine_assign");
}

// N-point weighted Procrustes, one workgroup per batch element (model_utils.py:343-436):
// w <- where(w < thresh, 0, w);  w <- w / (sum w + eps);  centroids;  H = sum (src - sc)^T (w (ref - rc));  R, t.
// Round 4: the workgroup's points stay in registers across the three passes (weight sum -> centroids -> correlation matrix; 8 points per
// thread at N = 2048), and the sums of a pass are reduced TOGETHER: a DPP prefix sum per value and wave (its last lane holds the wave
// total), one LDS exchange and one barrier pair per pass instead of one ds_bpermute butterfly + barrier pair per value (16 of them:
// 24 us for a few kiloflops).  Sums stay in double; only their association changes (1e-16 relative, far below the float results).
template <int NV>
__device__ __forceinline__ void block_sum_dv(double (&v)[NV], double (*red)[9]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_incl_scan_f64(v[k], lane);
  __syncthreads();  // (the previous pass has read `red`)
  if (lane == 63) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wave][k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
}

#define PR_PPT 8  // points per thread held in registers (N <= 256 * PR_PPT; longer clouds re-read)
__global__ __launch_bounds__(256) void procrustes_kernel(const float* __restrict__ src, const float* __restrict__ ref,
                                                         const float* __restrict__ wts, int N, float thresh, float eps,
                                                         float* __restrict__ Rout, float* __restrict__ tout) {
  __shared__ double red[4][9];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* S = src + (size_t)b * N * 3;
  const float* Q = ref + (size_t)b * N * 3;
  const float* W = wts ? wts + (size_t)b * N : nullptr;
  const bool cached = N <= 256 * PR_PPT;
  float cw[PR_PPT], cs[PR_PPT][3], cq[PR_PPT][3];
  if (cached) {
#pragma unroll
    for (int u = 0; u < PR_PPT; ++u) {
      const int i = t + 256 * u;
      const bool ok = i < N;
      float w = (ok && W) ? W[i] : (ok ? 1.0f : 0.0f);
      if (w < thresh) w = 0.f;
      cw[u] = w;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        cs[u][d] = ok ? S[i * 3 + d] : 0.f;
        cq[u][d] = ok ? Q[i * 3 + d] : 0.f;
      }
    }
  }
  double sw[1] = {0.0};
  if (cached) {
#pragma unroll
    for (int u = 0; u < PR_PPT; ++u) sw[0] += (double)cw[u];
  } else {
    for (int i = t; i < N; i += 256) {
      float w = W ? W[i] : 1.0f;
      if (w < thresh) w = 0.f;
      sw[0] += (double)w;
    }
  }
  block_sum_dv<1>(sw, red);
  const float den = (float)sw[0] + eps;
  double c[6] = {0, 0, 0, 0, 0, 0};
  if (cached) {
#pragma unroll
    for (int u = 0; u < PR_PPT; ++u) {
      const double wn = (double)(cw[u] / den);
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        c[d] += (double)cs[u][d] * wn;
        c[3 + d] += (double)cq[u][d] * wn;
      }
    }
  } else {
    for (int i = t; i < N; i += 256) {
      float w = W ? W[i] : 1.0f;
      if (w < thresh) w = 0.f;
      const double wn = (double)(w / den);
      for (int d = 0; d < 3; ++d) {
        c[d] += (double)S[i * 3 + d] * wn;
        c[3 + d] += (double)Q[i * 3 + d] * wn;
      }
    }
  }
  block_sum_dv<6>(c, red);
  double H[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (cached) {
#pragma unroll
    for (int u = 0; u < PR_PPT; ++u) {
      const double wn = (double)(cw[u] / den);
      double sd[3], qd[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        sd[d] = (double)cs[u][d] - c[d];
        qd[d] = wn * ((double)cq[u][d] - c[3 + d]);
      }
      // (a padding slot has weight 0: its q terms are exact zeros)
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int d = 0; d < 3; ++d) H[a * 3 + d] += sd[a] * qd[d];
    }
  } else {
    for (int i = t; i < N; i += 256) {
      float w = W ? W[i] : 1.0f;
      if (w < thresh) w = 0.f;
      const double wn = (double)(w / den);
      double sd[3], qd[3];
      for (int d = 0; d < 3; ++d) {
        sd[d] = (double)S[i * 3 + d] - c[d];
        qd[d] = wn * ((double)Q[i * 3 + d] - c[3 + d]);
      }
      for (int a = 0; a < 3; ++a)
        for (int d = 0; d < 3; ++d) H[a * 3 + d] += sd[a] * qd[d];
    }
  }
  block_sum_dv<9>(H, red);
  if (t == 0) {
    double R[9];
    rotation_from_H(H, R);
    for (int i = 0; i < 9; ++i) Rout[b * 9 + i] = (float)R[i];
    for (int i = 0; i < 3; ++i) tout[b * 3 + i] = (float)(c[3 + i] - (R[i * 3] * c[0] + R[i * 3 + 1] * c[1] + R[i * 3 + 2] * c[2]));
  }
}

extern "C" int sam6d_weighted_procrustes(const float* src, const float* ref, const float* weights, int B, int N, float weight_thresh,
                                         float eps, float* R, float* t, void* stream) {
  SAM6D_REQUIRE(src && ref && R && t && B >= 0 && N > 0, "weighted_procrustes: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(procrustes_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, src, ref, weights, N, weight_thresh, eps, R, t);
  SAM6D_LAUNCH_CHECK("weighted_procrustes");
}

// Fine pose score (model_utils.py:331-339):  dis_i = min_m |(p1_i - t) R - model_m|;  mask = [label1 > 0];
// score = sum([dis < thr] * mask) / (sum(mask) + 1e-8) * mean(mask);  t_out = t * (radius + 1e-6) (fine_point_matching.py:78)
// Round 4: a workgroup owns 64 points and its four waves each scan a quarter of the CAD points (the minimum does not depend on the order;
// was: 256 points per workgroup, every thread walking all P model points alone -- one wave per SIMD on a 1024-step chain, 31 us).
__global__ __launch_bounds__(256) void fine_near_kernel(const float* __restrict__ pts1, const float* __restrict__ R,
                                                        const float* __restrict__ t, const float* __restrict__ model,
                                                        const float* __restrict__ radius, const int* __restrict__ label1, int N,
                                                        int P, float thr, float* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  __shared__ float part[4][64];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float den = radius[b] + 1e-6f;
  const float* mb = model + (size_t)b * P * 3;
  for (int i = tid; i < P; i += 256) {
    const float x = mb[i * 3] / den, y = mb[i * 3 + 1] / den, z = mb[i * 3 + 2] / den;
    sm[i * 4] = x; sm[i * 4 + 1] = y; sm[i * 4 + 2] = z; sm[i * 4 + 3] = sqnorm3(x, y, z);
  }
  __syncthreads();
  const int i = blockIdx.x * 64 + lane;
  float mn = INFINITY;
  if (i < N) {
    const float* Rb = R + b * 9;
    const float* p = pts1 + ((size_t)b * N + i) * 3;
    const float d0 = p[0] - t[b * 3], d1 = p[1] - t[b * 3 + 1], d2 = p[2] - t[b * 3 + 2];
    const float x0 = fmaf(d2, Rb[6], fmaf(d1, Rb[3], d0 * Rb[0]));
    const float x1 = fmaf(d2, Rb[7], fmaf(d1, Rb[4], d0 * Rb[1]));
    const float x2 = fmaf(d2, Rb[8], fmaf(d1, Rb[5], d0 * Rb[2]));
    const float sx = sqnorm3(x0, x1, x2);
    const int chunk = (P + 3) >> 2, m0 = wave * chunk, m1 = min(P, m0 + chunk);
    for (int m = m0; m < m1; ++m) {
      const float4 q = *reinterpret_cast<const float4*>(&sm[m * 4]);
      mn = fminf(mn, pdist3(x0, x1, x2, sx, q.x, q.y, q.z, q.w));
    }
  }
  part[wave][lane] = mn;
  __syncthreads();
  if (wave == 0) {
    float near = 0.f, mk = 0.f;
    if (i < N) {
      mn = fminf(fminf(part[0][lane], part[1][lane]), fminf(part[2][lane], part[3][lane]));
      mk = label1[(size_t)b * N + i] > 0 ? 1.f : 0.f;
      near = (sqrtf(mn) < thr) ? mk : 0.f;
    }
    near = wave_sum_dpp(near);  // integer-valued partial sums: exact in fp32 whatever the order
    mk = wave_sum_dpp(mk);
    if (lane == 0) {
      atomicAdd(&cnt[b * 2], near);
      atomicAdd(&cnt[b * 2 + 1], mk);
    }
  }
}

// The same count on the fp32 matrix cores, in the arithmetic of score_hyp_mfma_kernel (two v_mfma_f32_32x32x2_f32 on a zero accumulator
// give rn(|x|^2 - 2 xy) in the fma order of pdist3, the vector ALU adds |y|^2 and keeps the running minimum: the same bits as
// fine_near_kernel, whose thread walks the P CAD points with ~8 vector instructions per pair -- 31 us for 32 x 2048 x 1024 pairs).
// A wave owns FN_CT column tiles of 32 scene points and walks the 32-row tiles of the CAD points.
#define FN_CT 2
__global__ __launch_bounds__(256) void fine_near_mfma_kernel(const float* __restrict__ pts1, const float* __restrict__ R,
                                                             const float* __restrict__ t, const float* __restrict__ model,
                                                             const float* __restrict__ radius, const int* __restrict__ label1, int N,
                                                             int P, int Ppad, float thr, float* __restrict__ cnt) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // [4][Ppad]: -2 y0 | -2 y1 | -2 y2 | 1;  then [Ppad]: |y|^2 (+inf padding)
  __shared__ float red[2][4];
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, kk = lane >> 5;
  const float den = radius[b] + 1e-6f;
  const float* mb = model + (size_t)b * P * 3;
  for (int i = tid; i < Ppad; i += 256) {
    float x = 0.f, y = 0.f, z = 0.f, sq = INFINITY;
    if (i < P) {
      x = mb[i * 3] / den; y = mb[i * 3 + 1] / den; z = mb[i * 3 + 2] / den;
      sq = sqnorm3(x, y, z);
    }
    sm[i] = -2.0f * x; sm[Ppad + i] = -2.0f * y; sm[2 * Ppad + i] = -2.0f * z; sm[3 * Ppad + i] = 1.0f;
    sm[4 * Ppad + i] = sq;
  }
  const int e0 = (blockIdx.x * 4 + wave) * FN_CT * 32 + j;
  const float* Rb = R + b * 9;
  float b1[FN_CT], b2[FN_CT], mn[FN_CT];
#pragma unroll
  for (int c = 0; c < FN_CT; ++c) {
    const int i = e0 + 32 * c;
    float x0 = 0.f, x1 = 0.f, x2 = 0.f, sx = 0.f;
    if (i < N) {
      const float* p = pts1 + ((size_t)b * N + i) * 3;
      const float d0 = p[0] - t[b * 3], d1 = p[1] - t[b * 3 + 1], d2 = p[2] - t[b * 3 + 2];
      x0 = fmaf(d2, Rb[6], fmaf(d1, Rb[3], d0 * Rb[0]));
      x1 = fmaf(d2, Rb[7], fmaf(d1, Rb[4], d0 * Rb[1]));
      x2 = fmaf(d2, Rb[8], fmaf(d1, Rb[5], d0 * Rb[2]));
      sx = sqnorm3(x0, x1, x2);
    }
    b1[c] = kk ? x1 : x0;
    b2[c] = kk ? sx : x2;
    mn[c] = INFINITY;
  }
  __syncthreads();
  f32x16 zero;
#pragma unroll
  for (int v = 0; v < 16; ++v) zero[v] = 0.f;
  const float* a1p = sm + kk * Ppad + j;
  const float* a2p = sm + (2 + kk) * Ppad + j;
  const float* syp = sm + 4 * Ppad + kk * 4;
  for (int r = 0; r < Ppad; r += 32) {
    const float a1 = a1p[r], a2 = a2p[r];
    const float4 s0 = *reinterpret_cast<const float4*>(syp + r), s1 = *reinterpret_cast<const float4*>(syp + r + 8);
    const float4 s2 = *reinterpret_cast<const float4*>(syp + r + 16), s3 = *reinterpret_cast<const float4*>(syp + r + 24);
#pragma unroll
    for (int c = 0; c < FN_CT; ++c) {
      f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1[c], zero, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, b2[c], acc, 0, 0, 0);
      // accumulator register v of this lane: CAD row r + 8 (v >> 2) + 4 kk + (v & 3), column j
      const float m0 = fminf(fminf(acc[0] + s0.x, acc[1] + s0.y), fminf(acc[2] + s0.z, acc[3] + s0.w));
      const float m1 = fminf(fminf(acc[4] + s1.x, acc[5] + s1.y), fminf(acc[6] + s1.z, acc[7] + s1.w));
      const float m2 = fminf(fminf(acc[8] + s2.x, acc[9] + s2.y), fminf(acc[10] + s2.z, acc[11] + s2.w));
      const float m3 = fminf(fminf(acc[12] + s3.x, acc[13] + s3.y), fminf(acc[14] + s3.z, acc[15] + s3.w));
      mn[c] = fminf(fminf(mn[c], m0), fminf(fminf(m1, m2), m3));
    }
  }
  float near = 0.f, mk = 0.f;
#pragma unroll
  for (int c = 0; c < FN_CT; ++c) {
    float m = fminf(mn[c], __shfl_xor(mn[c], 32, 64));
    m = m < 0.0f ? 0.0f : m;  // pdist3's clamp (it commutes with the minimum)
    const int i = e0 + 32 * c;
    if (kk == 0 && i < N) {
      const float k1 = label1[(size_t)b * N + i] > 0 ? 1.f : 0.f;
      mk += k1;
      near += (sqrtf(m) < thr) ? k1 : 0.f;
    }
  }
  near = wave_sum_dpp(near);  // integer-valued partial sums: exact in fp32 whatever the order
  mk = wave_sum_dpp(mk);
  if (lane == 0) { red[0][wave] = near; red[1][wave] = mk; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(&cnt[b * 2], (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]));
    atomicAdd(&cnt[b * 2 + 1], (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]));
  }
}

__global__ void zero_kernel(float* p, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ void fine_finish_kernel(const float* __restrict__ cnt, const float* __restrict__ radius, int B, int N,
                                   float* __restrict__ t, float* __restrict__ score) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  const float near = cnt[b * 2], mk = cnt[b * 2 + 1];
  score[b] = (near / (mk + 1e-8f)) * (mk / (float)N);
  const float s = radius[b] + 1e-6f;
  t[b * 3] *= s; t[b * 3 + 1] *= s; t[b * 3 + 2] *= s;
}

extern "C" int sam6d_fine_score(const float* pts1, const float* R, float* t, const float* model, const float* radius,
                                const int* label1, int B, int N, int P, float dis_thres, float* cnt_ws, float* score, void* stream) {
  SAM6D_REQUIRE(pts1 && R && t && model && radius && label1 && cnt_ws && score, "fine_score: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && P > 0 && P <= 8192 && B <= 65535, "fine_score: bad sizes (P <= 8192)");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  static unsigned long long fn_done = 0;
  if (sam6d_first_use_on_device(&fn_done)) {  // P = 8192 CAD points are 128 KB of dynamic LDS (the default limit is 64 KB)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fine_near_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8192 * 16);
    if (e != hipSuccess) {
      sam6d_set_error("fine_score: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&fn_done);
  }
  hipLaunchKernelGGL(zero_kernel, dim3(cdiv(2 * B, 256)), dim3(256), 0, s, cnt_ws, 2 * B);
  const char* fn_env = getenv("SAM6D_FINE_NEAR_MFMA");  // A/B switch, read per call (one call per step): 0 = the vector-ALU kernel
  const int use_mfma = (fn_env && fn_env[0] == '0') ? 0 : 1;
  const int Ppad = (P + 31) & ~31;
  if (use_mfma && (size_t)Ppad * 20 <= 65536 - 64)
    hipLaunchKernelGGL(fine_near_mfma_kernel, dim3(cdiv(N, 4 * FN_CT * 32), B), dim3(256), (size_t)Ppad * 20, s, pts1, R, t, model, radius,
                       label1, N, P, Ppad, dis_thres, cnt_ws);
  else
    hipLaunchKernelGGL(fine_near_kernel, dim3(cdiv(N, 64), B), dim3(256), (size_t)P * 16, s, pts1, R, t, model, radius, label1, N, P,
                       dis_thres, cnt_ws);
  hipLaunchKernelGGL(fine_finish_kernel, dim3(cdiv(B, 256)), dim3(256), 0, s, cnt_ws, radius, B, N, t, score);
  SAM6D_LAUNCH_CHECK("fine_score");
}

// =========================================================================================================
// pairwise_distance (model_utils.py:101-128) as a stand-alone entry: the K = 3 squared distances in the torch-CPU bit recipe.
// The path's own kernels (geo_knn / geo_index, score_hyp, fine_near) inline the same pdist3(); this entry exposes the recipe at the
// reference's call signature so that it is checked directly against tests/golden/pairwise.npz.
// =========================================================================================================
__global__ __launch_bounds__(256) void pairwise_distance_kernel(const float* __restrict__ x, const float* __restrict__ y, int N, int M,
                                                                float* __restrict__ out) {
  const int b = blockIdx.z;
  const int n = blockIdx.y;
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  const float* xp = x + ((size_t)b * N + n) * 3;
  const float* yp = y + ((size_t)b * M + m) * 3;
  const float x0 = xp[0], x1 = xp[1], x2 = xp[2], y0 = yp[0], y1 = yp[1], y2 = yp[2];
  out[((size_t)b * N + n) * M + m] = pdist3(x0, x1, x2, sqnorm3(x0, x1, x2), y0, y1, y2, sqnorm3(y0, y1, y2));
}

extern "C" int sam6d_pairwise_distance(const float* x, const float* y, int B, int N, int M, float* out, void* stream) {
  SAM6D_REQUIRE(x && y && out, "pairwise_distance: null pointer");
  SAM6D_REQUIRE(B >= 0 && B <= 65535 && N >= 0 && N <= 65535 && M >= 0, "pairwise_distance: bad sizes (B, N <= 65535)");
  if (B == 0 || N == 0 || M == 0) return 0;
  hipLaunchKernelGGL(pairwise_distance_kernel, dim3(cdiv(M, 256), N, B), dim3(256), 0, (hipStream_t)stream, x, y, N, M, out);
  SAM6D_LAUNCH_CHECK("pairwise_distance");
}
