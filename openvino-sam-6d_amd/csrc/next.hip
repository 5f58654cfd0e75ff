// Rows SURVEY 8f marks "next" (the callers either side of the matching path), built to the same bar:
//   radius normalisation of ViTEncoder.forward          PEM/model/feature_extraction.py:128-139
//   masked patch-descriptor post-processing of DINOv2   ISM/model/dinov2.py:258-270, 308-326
//   depth back-projection of get_test_data              PEM/utils/data_utils.py:92-110
// (the third "next" item, FPS over the 210 000-point template cloud, is fps_big_kernel in pointops.hip)
#include "common.h"
#include "../../include/sam6d_hip.h"

// radius[b] = max_n |dense_po[b,n]|   (torch.norm(dim=2).max(1)[0]; norm = sqrt(fma(z,z, fma(y,y, x*x))), the
// torch-CPU vector-norm recipe also used in geo.hip).  One workgroup per cloud.
__global__ __launch_bounds__(256) void radius_kernel(const float* __restrict__ po, int N, float* __restrict__ radius) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* p = po + (size_t)b * N * 3;
  float m = 0.f;  // norms are >= 0
  for (int i = t; i < N; i += 256) {
    const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    m = fmaxf(m, sqrtf(fmaf(z, z, fmaf(y, y, x * x))));
  }
  m = wave_max(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  if (t == 0) radius[b] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// y[b, i, :] = x[b, i, :] / (radius[b] + 1e-6)
__global__ __launch_bounds__(256) void scale_by_radius_kernel(const float* __restrict__ x, const float* __restrict__ radius,
                                                              long per_b, long total, float* __restrict__ y) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  y[e] = x[e] / (radius[e / per_b] + 1e-6f);
}

extern "C" int sam6d_radius_normalize(const float* dense_po, const float* pts, int B, int Npo, int Npm, float* radius,
                                      float* po_out, float* pm_out, void* stream) {
  SAM6D_REQUIRE(dense_po && pts && radius && po_out && pm_out, "radius_normalize: null pointer");
  SAM6D_REQUIRE(B >= 0 && Npo > 0 && Npm > 0, "radius_normalize: bad sizes");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(radius_kernel, dim3(B), dim3(256), 0, s, dense_po, Npo, radius);
  const long t1 = (long)B * Npo * 3, t2 = (long)B * Npm * 3;
  hipLaunchKernelGGL(scale_by_radius_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, dense_po, radius,
                     (long)Npo * 3, t1, po_out);
  hipLaunchKernelGGL(scale_by_radius_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, pts, radius,
                     (long)Npm * 3, t2, pm_out);
  SAM6D_LAUNCH_CHECK("radius_normalize");
}

// out[n,p,:] = normalize(feats[n,p,:] * [avgpool_patch(mask[n])[p] > thresh])   -- one wave per (image, patch)
__global__ __launch_bounds__(256) void masked_patch_kernel(const float* __restrict__ feats, const float* __restrict__ masks,
                                                           int P, int D, int H, int W, int patch, float thresh, long total,
                                                           float* __restrict__ out) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (w >= total) return;
  const int lane = threadIdx.x & 63;
  const long n = w / P;
  const int p = (int)(w % P);
  const int pw = W / patch;
  const int py = p / pw, px = p % pw;
  const float* m = masks + (size_t)n * H * W + (size_t)py * patch * W + px * patch;
  float s = 0.f;
  for (int e = lane; e < patch * patch; e += 64) s += m[(e / patch) * W + (e % patch)];
  s = wave_sum(s);
  const bool keep = (s / (float)(patch * patch)) > thresh;  // nn.AvgPool2d(patch)(mask) > validpatch_thresh
  const float* f = feats + (size_t)w * D;
  float* o = out + (size_t)w * D;
  float ss = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(f + c);
    ss += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
  }
  const float nrm = fmaxf(sqrtf(wave_sum(ss)), 1e-12f);
  for (int c = lane * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(f + c);
    *reinterpret_cast<float4*>(o + c) =
        keep ? make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}

extern "C" int sam6d_masked_patch_normalize(const float* feats, const float* masks, int N, int P, int D, int H, int W,
                                            int patch, float thresh, float* out, void* stream) {
  SAM6D_REQUIRE(feats && masks && out, "masked_patch_normalize: null pointer");
  SAM6D_REQUIRE(N >= 0 && P > 0 && D > 0 && (D & 3) == 0 && patch > 0 && H % patch == 0 && W % patch == 0 &&
                    (H / patch) * (W / patch) == P,
                "masked_patch_normalize: need D %% 4 == 0 and P == (H/patch)*(W/patch)");
  const long total = (long)N * P;
  if (total == 0) return 0;
  hipLaunchKernelGGL(masked_patch_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, feats, masks, P,
                     D, H, W, patch, thresh, total, out);
  SAM6D_LAUNCH_CHECK("masked_patch_normalize");
}

// cloud[r, c, :] = ((c0 + c - cx) * z / fx, (r0 + r - cy) * z / fy, z), z = depth[r0 + r, c0 + c]   (fp32, numpy order:
// subtract, multiply, divide -- get_point_cloud_from_depth with the crop bbox = [r0, r1, c0, c1])
__global__ __launch_bounds__(256) void depth_to_cloud_kernel(const float* __restrict__ depth, int W, int r0, int c0, int h, int w,
                                                             float fx, float fy, float cx, float cy, float* __restrict__ cloud) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)h * w) return;
  const int r = (int)(e / w) + r0, c = (int)(e % w) + c0;
  const float z = depth[(size_t)r * W + c];
  cloud[e * 3 + 0] = ((float)c - cx) * z / fx;
  cloud[e * 3 + 1] = ((float)r - cy) * z / fy;
  cloud[e * 3 + 2] = z;
}

extern "C" int sam6d_depth_to_cloud(const float* depth, int H, int W, int r0, int r1, int c0, int c1, float fx, float fy, float cx,
                                    float cy, float* cloud, void* stream) {
  SAM6D_REQUIRE(depth && cloud, "depth_to_cloud: null pointer");
  SAM6D_REQUIRE(H > 0 && W > 0 && 0 <= r0 && r0 <= r1 && r1 <= H && 0 <= c0 && c0 <= c1 && c1 <= W,
                "depth_to_cloud: bbox [%d,%d)x[%d,%d) outside the %dx%d depth map", r0, r1, c0, c1, H, W);
  const long total = (long)(r1 - r0) * (c1 - c0);
  if (total == 0) return 0;
  hipLaunchKernelGGL(depth_to_cloud_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, depth, W,
                     r0, c0, r1 - r0, c1 - c0, fx, fy, cx, cy, cloud);
  SAM6D_LAUNCH_CHECK("depth_to_cloud");
}
