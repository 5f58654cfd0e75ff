// Proposal geometry of get_test_data (PEM/run_inference_custom_pytorch.py:316-355) on a resident depth map
// and resident proposal masks: mask & depth > 0, square crop box (get_bbox, PEM/utils/data_utils.py:125-160), masked pixels of
// the crop in row-major order, back-projected points (get_point_cloud_from_depth, :92-110), radius filter around their mean,
// the caller-supplied random choice, and the indices of the chosen pixels in the resized crop (get_resize_rgb_choose, :113-123).
// HBM-bound integer / byte work: one workgroup per proposal, ballot-prefix compaction keeps the reference's pixel order.
#include "common.h"
#include "../../include/sam6d_hip.h"

// ---- 1. valid-pixel count and crop box --------------------------------------------------------------------------------
// bbox[i] = {rmin, rmax, cmin, cmax} exactly as get_bbox computes it from (mask > 0) & (depth > 0); count[i] = number of
// valid pixels (the caller skips proposals with count <= 32, run_inference_custom_pytorch.py:320-324).
__global__ __launch_bounds__(256) void mask_bbox_kernel(const unsigned char* __restrict__ masks, const float* __restrict__ depth, int H,
                                                        int W, int* __restrict__ bbox, int* __restrict__ count) {
  __shared__ int s_rmin, s_rmax, s_cmin, s_cmax, s_cnt;
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (t == 0) {
    s_rmin = H;
    s_rmax = -1;
    s_cmin = W;
    s_cmax = -1;
    s_cnt = 0;
  }
  __syncthreads();
  const unsigned char* m = masks + (size_t)i * H * W;
  int rmin = H, rmax = -1, cmin = W, cmax = -1, cnt = 0;
  for (int r = wave; r < H; r += 4) {  // one wave per image row, lanes over columns (coalesced)
    for (int c0 = 0; c0 < W; c0 += 64) {
      const int c = c0 + lane;
      const bool v = c < W && m[(size_t)r * W + c] != 0 && depth[(size_t)r * W + c] > 0.f;
      if (v) {
        cmin = min(cmin, c);
        cmax = max(cmax, c);
        rmin = min(rmin, r);
        rmax = max(rmax, r);
        ++cnt;
      }
    }
  }
  atomicMin(&s_rmin, rmin);
  atomicMax(&s_rmax, rmax);
  atomicMin(&s_cmin, cmin);
  atomicMax(&s_cmax, cmax);
  atomicAdd(&s_cnt, cnt);
  __syncthreads();
  if (t == 0) {
    count[i] = s_cnt;
    int r0 = s_rmin, r1 = s_rmax + 1, c0 = s_cmin, c1 = s_cmax + 1;
    if (s_cnt == 0) {
      r0 = r1 = c0 = c1 = 0;
    } else {
      // get_bbox: a square of side min(max(r_b, c_b), min(H, W)) around the integer centre, shifted back into the image
      const int rb = r1 - r0, cb = c1 - c0;
      const int b = min(max(rb, cb), min(H, W));
      const int cr = (r0 + r1) / 2, cc = (c0 + c1) / 2;  // int((a + b) / 2) for non-negative a + b
      const int half = b / 2;                             // int(b / 2)
      r0 = cr - half;
      r1 = cr + half;
      c0 = cc - half;
      c1 = cc + half;
      if (r0 < 0) {
        r1 += -r0;
        r0 = 0;
      }
      if (c0 < 0) {
        c1 += -c0;
        c0 = 0;
      }
      if (r1 > H) {
        r0 -= r1 - H;
        r1 = H;
      }
      if (c1 > W) {
        c0 -= c1 - W;
        c1 = W;
      }
    }
    bbox[i * 4 + 0] = r0;
    bbox[i * 4 + 1] = r1;
    bbox[i * 4 + 2] = c0;
    bbox[i * 4 + 3] = c1;
  }
}

extern "C" int sam6d_mask_bbox(const unsigned char* masks, const float* depth, int N, int H, int W, int* bbox, int* count,
                               void* stream) {
  SAM6D_REQUIRE(masks && depth && bbox && count, "mask_bbox: null pointer");
  SAM6D_REQUIRE(N >= 0 && H > 0 && W > 0, "mask_bbox: bad sizes");
  if (N == 0) return 0;
  hipLaunchKernelGGL(mask_bbox_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, masks, depth, H, W, bbox, count);
  SAM6D_LAUNCH_CHECK("mask_bbox");
}

// ---- 2. masked pixels of the crop, in row-major order, with their back-projected points -------------------------------
// choose[i, k] = flat index (row * crop_w + col) of the k-th valid pixel inside bbox[i]; cloud[i, k] = its 3-D point in the
// get_point_cloud_from_depth recipe ((col - cx) * z / fx, (row - cy) * z / fy, z; fp32, in that order).  n_valid[i] = how many.
__global__ __launch_bounds__(1024) void crop_points_kernel(const unsigned char* __restrict__ masks, const float* __restrict__ depth,
                                                           int H, int W, const int* __restrict__ bbox, float fx, float fy, float cx,
                                                           float cy, int cap, int* __restrict__ choose, float* __restrict__ cloud,
                                                           int* __restrict__ n_valid) {
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int r0 = bbox[i * 4], r1 = bbox[i * 4 + 1], c0 = bbox[i * 4 + 2], c1 = bbox[i * 4 + 3];
  const int ch = r1 - r0, cw = c1 - c0;
  const long npix = (long)ch * cw;
  const unsigned char* m = masks + (size_t)i * H * W;
  if (t == 0) s_base = 0;
  __syncthreads();
  for (long p0 = 0; p0 < npix; p0 += 1024) {
    const long p = p0 + t;
    bool v = false;
    int r = 0, c = 0;
    float z = 0.f;
    if (p < npix) {
      r = r0 + (int)(p / cw);
      c = c0 + (int)(p % cw);
      z = depth[(size_t)r * W + c];
      v = m[(size_t)r * W + c] != 0 && z > 0.f;
    }
    const unsigned long long bal = __ballot(v);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (v) {
      const int k = off + __popcll(bal & ((1ull << lane) - 1ull));
      if (k < cap) {
        choose[(size_t)i * cap + k] = (int)p;
        float* o = cloud + ((size_t)i * cap + k) * 3;
        o[0] = ((float)c - cx) * z / fx;
        o[1] = ((float)r - cy) * z / fy;
        o[2] = z;
      }
    }
    __syncthreads();
    if (t == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (t == 0) n_valid[i] = min(s_base, cap);
}

// ---- 3. radius filter around the mean point ---------------------------------------------------------------------------
// center = np.mean(cloud, axis=0) of an (n, 3) float32 array = sequential fp32 sum over the rows, then / n; keep the points with
// sqrt((dx*dx + dy*dy) + dz*dz) < thr (np.linalg.norm(axis=1) of float32), thr = fp32(radius) * fp32(1.2).  In-place compaction
// of (choose, cloud) in order; n_keep[i] = survivors (the caller skips proposals with fewer than 4).
__global__ __launch_bounds__(1024) void radius_filter_kernel(int cap, const int* __restrict__ n_valid, float thr, int* __restrict__ choose,
                                                             float* __restrict__ cloud, int* __restrict__ n_keep,
                                                             float* __restrict__ center_out) {
  __shared__ float s_c[3];
  __shared__ int s_wave[16];
  __shared__ int s_base;
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int n = n_valid[i];
  int* ch = choose + (size_t)i * cap;
  float* cl = cloud + (size_t)i * cap * 3;
  if (t < 3) {  // three threads, one coordinate each: the reference's summation order
    float s = 0.f;
    for (int k = 0; k < n; ++k) s += cl[k * 3 + t];
    s_c[t] = n > 0 ? s / (float)n : 0.f;
    center_out[i * 3 + t] = s_c[t];
  }
  if (t == 0) s_base = 0;
  __syncthreads();
  const float cx = s_c[0], cy = s_c[1], cz = s_c[2];
  for (int k0 = 0; k0 < n; k0 += 1024) {
    const int k = k0 + t;
    bool v = false;
    int cv = 0;
    float x = 0.f, y = 0.f, z = 0.f;
    if (k < n) {
      cv = ch[k];
      x = cl[k * 3];
      y = cl[k * 3 + 1];
      z = cl[k * 3 + 2];
      const float dx = x - cx, dy = y - cy, dz = z - cz;
      v = sqrtf((dx * dx + dy * dy) + dz * dz) < thr;
    }
    const unsigned long long bal = __ballot(v);
    if (lane == 0) s_wave[wave] = __popcll(bal);
    __syncthreads();  // every thread has read its element of this chunk: the in-place writes below land at indices <= k
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    if (v) {
      const int d = off + __popcll(bal & ((1ull << lane) - 1ull));
      ch[d] = cv;
      cl[d * 3] = x;
      cl[d * 3 + 1] = y;
      cl[d * 3 + 2] = z;
    }
    __syncthreads();
    if (t == 0) {
      int tot = 0;
      for (int w = 0; w < 16; ++w) tot += s_wave[w];
      s_base += tot;
    }
    __syncthreads();
  }
  if (t == 0) n_keep[i] = s_base;
}

extern "C" int sam6d_crop_masked_points(const unsigned char* masks, const float* depth, int N, int H, int W, const int* bbox, float fx,
                                        float fy, float cx, float cy, int cap, int* choose, float* cloud, int* n_valid,
                                        void* stream) {
  SAM6D_REQUIRE(masks && depth && bbox && choose && cloud && n_valid, "crop_masked_points: null pointer");
  SAM6D_REQUIRE(N >= 0 && H > 0 && W > 0 && cap > 0, "crop_masked_points: bad sizes");
  if (N == 0) return 0;
  hipLaunchKernelGGL(crop_points_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, masks, depth, H, W, bbox, fx, fy, cx, cy, cap,
                     choose, cloud, n_valid);
  SAM6D_LAUNCH_CHECK("crop_masked_points");
}

extern "C" int sam6d_radius_filter(int N, int cap, const int* n_valid, float radius, int* choose, float* cloud, int* n_keep,
                                   float* center, void* stream) {
  SAM6D_REQUIRE(n_valid && choose && cloud && n_keep && center, "radius_filter: null pointer");
  SAM6D_REQUIRE(N >= 0 && cap > 0, "radius_filter: bad sizes");
  if (N == 0) return 0;
  // `radius * 1.2` with radius an np.float32 scalar is a float64 product under numpy 1.26 (the version the reference pins); the
  // float32 norms are then compared with its float32 value
  const float thr = (float)((double)radius * 1.2);
  hipLaunchKernelGGL(radius_filter_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, cap, n_valid, thr, choose, cloud, n_keep,
                     center);
  SAM6D_LAUNCH_CHECK("radius_filter");
}

// ---- 4. the random choice (drawn by the caller) and the pixel indices in the resized crop ------------------------------
// pts[i, j] = cloud[i, sel[i, j]]; rgb_choose[i, j] = floor(row * (S / crop_h)) * S + floor(col * (S / crop_w)) with row / col of
// choose[i, sel[i, j]] in the crop, in float64 like numpy (get_resize_rgb_choose)
__global__ __launch_bounds__(256) void choose_points_kernel(int cap, const int* __restrict__ choose, const float* __restrict__ cloud,
                                                            const int* __restrict__ bbox, const int* __restrict__ sel, int ns,
                                                            int img_size, long total, float* __restrict__ pts,
                                                            long long* __restrict__ rgb_choose) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int i = (int)(e / ns);
  const int k = min(max(sel[e], 0), cap - 1);  // the caller draws sel in [0, n_keep); never index outside the buffers
  const int cv = choose[(size_t)i * cap + k];
  const float* c = cloud + ((size_t)i * cap + k) * 3;
  pts[e * 3] = c[0];
  pts[e * 3 + 1] = c[1];
  pts[e * 3 + 2] = c[2];
  const int chh = bbox[i * 4 + 1] - bbox[i * 4], cww = bbox[i * 4 + 3] - bbox[i * 4 + 2];
  const double rh = (double)img_size / (double)chh, rw = (double)img_size / (double)cww;
  const long long row = cv / cww, col = cv % cww;
  rgb_choose[e] = (long long)(floor((double)row * rh) * (double)img_size + floor((double)col * rw));
}

extern "C" int sam6d_choose_points(int N, int cap, const int* choose, const float* cloud, const int* bbox, const int* sel, int ns,
                                   int img_size, float* pts, long long* rgb_choose, void* stream) {
  SAM6D_REQUIRE(choose && cloud && bbox && sel && pts && rgb_choose, "choose_points: null pointer");
  SAM6D_REQUIRE(N >= 0 && cap > 0 && ns > 0 && img_size > 0, "choose_points: bad sizes");
  const long total = (long)N * ns;
  if (total == 0) return 0;
  hipLaunchKernelGGL(choose_points_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, cap, choose, cloud,
                     bbox, sel, ns, img_size, total, pts, rgb_choose);
  SAM6D_LAUNCH_CHECK("choose_points");
}
