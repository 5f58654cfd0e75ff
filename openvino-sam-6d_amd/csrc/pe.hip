// Helpers of the fine-matching PositionalEncoding (PEM/model/fine_point_matching.py:102-144):
//   QueryAndGroup (PEM/model/pointnet2/pointnet2_utils.py:326-403) as row-major 6-vectors for the MLP GEMMs, the max
//   over each ball (torch.amax(dim=3), fine_point_matching.py:131,139), and small row utilities of the dense path.
// The 1x1-conv SharedMLP layers themselves run on the matrix cores through gemm_nt (BatchNorm folded to a per-column
// scale/shift in the GEMM epilogue).
#include "common.h"
#include "../../include/sam6d_hip.h"

// rows[(b*M + j)*S + s] = { p[idx]-q_j (3), p[idx] (3) },  q_j = pts_j + 1e-8f  (the reference's new_xyz, :117)
__global__ __launch_bounds__(256) void pe_group_rows_kernel(const float* __restrict__ pts, const int* __restrict__ idx,
                                                            int N, int S, long total, float* __restrict__ rows) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long bj = e / S;
  const long b = bj / N;
  const int a = idx[e];
  const float* pb = pts + b * N * 3;
  const float* pq = pts + bj * 3;
  const bool ok = a >= 0 && a < N;
  const float x = ok ? pb[a * 3] : 0.f, y = ok ? pb[a * 3 + 1] : 0.f, z = ok ? pb[a * 3 + 2] : 0.f;
  const float qx = pq[0] + 0.00000001f, qy = pq[1] + 0.00000001f, qz = pq[2] + 0.00000001f;
  float* o = rows + e * 6;
  o[0] = x - qx;
  o[1] = y - qy;
  o[2] = z - qz;
  o[3] = x;
  o[4] = y;
  o[5] = z;
}

extern "C" int sam6d_pe_group_rows(const float* pts, const int* idx, int B, int N, int S, float* rows, void* stream) {
  SAM6D_REQUIRE(pts && idx && rows && B >= 0 && N > 0 && S > 0, "pe_group_rows: bad arguments");
  const long total = (long)B * N * S;
  if (total == 0) return 0;
  hipLaunchKernelGGL(pe_group_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pts,
                     idx, N, S, total, rows);
  SAM6D_LAUNCH_CHECK("pe_group_rows");
}

// out[g, off + c] = max_{s < S} x[(g*S + s), c],  c < C (C % 4 == 0).  One thread per (group, 4 channels).
__global__ __launch_bounds__(256) void group_max_kernel(const float* __restrict__ x, int S, int C, long groups, long ldo,
                                                        int off, float* __restrict__ out) {
  const int c4 = C >> 2;
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= groups * c4) return;
  const long g = e / c4;
  const int c = (int)(e % c4) * 4;
  const float* p = x + (g * S) * C + c;
  float4 m = *reinterpret_cast<const float4*>(p);
  for (int s = 1; s < S; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long)s * C);
    m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
  }
  *reinterpret_cast<float4*>(out + g * ldo + off + c) = m;
}

extern "C" int sam6d_group_max(const float* x, long groups, int S, int C, long ldo, int off, float* out, void* stream) {
  SAM6D_REQUIRE(x && out && groups >= 0 && S > 0 && C > 0 && (C & 3) == 0 && (ldo & 3) == 0 && (off & 3) == 0,
                "group_max: bad arguments");
  if (groups == 0) return 0;
  const long n = groups * (C >> 2);
  hipLaunchKernelGGL(group_max_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, S, C, groups,
                     ldo, off, out);
  SAM6D_LAUNCH_CHECK("group_max");
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
