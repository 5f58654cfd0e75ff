// Detection bookkeeping either side of the ISM scoring path (SURVEY 8f "next"): the tensor work of
// ISM/model/utils.py `Detections` -- remove_very_small_detections (:96-105), apply_nms_per_object_id (:107-119),
// apply_nms (:121-126), filter (:188-190).  torchvision.ops.nms (torchvision 0.15/0.20, pinned by the reference's
// environment files but not vendored) is restated from its published kernel: stable descending sort by score, greedy
// suppression of every later box with IoU > thresh, IoU = inter / (area_a + area_b - inter), survivors returned in
// score order.  HBM-bound integer/byte work: no MFMA here.
#include "common.h"
#include "../../include/sam6d_hip.h"

// keep[i] = box_area(boxes[i]) / (H*W) > thr_box  &&  masks[i].sum() / (H*W) > thr_mask     (one workgroup per detection)
// boxes are int64 xyxy (Detections.__init__ casts them with .long()); the int64 area is converted to fp32 before the
// division exactly as torch's true_divide does.  Mask sums of binary masks are integers < 2^24, exact in any order.
__global__ __launch_bounds__(256) void small_keep_kernel(const long long* __restrict__ boxes, const float* __restrict__ masks,
                                                         long HW, float thr_box, float thr_mask,
                                                         unsigned char* __restrict__ keep) {
  __shared__ float red[4];
  const int i = blockIdx.x, t = threadIdx.x;
  const float* m = masks + (size_t)i * HW;
  float s = 0.f;
  if ((HW & 3) == 0) {
    const float4* m4 = reinterpret_cast<const float4*>(m);
    for (long e = t; e < HW / 4; e += 256) {
      const float4 v = m4[e];
      s += (v.x + v.y) + (v.z + v.w);
    }
  } else {
    for (long e = t; e < HW; e += 256) s += m[e];
  }
  s = wave_sum(s);
  if ((t & 63) == 0) red[t >> 6] = s;
  __syncthreads();
  if (t == 0) {
    const float msum = (red[0] + red[1]) + (red[2] + red[3]);
    const long long* b = boxes + (size_t)i * 4;
    const long long area = (b[2] - b[0]) * (b[3] - b[1]);
    const float img = (float)HW;
    keep[i] = ((float)area / img > thr_box) && (msum / img > thr_mask);
  }
}

extern "C" int sam6d_detections_small_keep(const long long* boxes, const float* masks, int N, int H, int W, float thr_box,
                                           float thr_mask, unsigned char* keep, void* stream) {
  SAM6D_REQUIRE(boxes && masks && keep, "detections_small_keep: null pointer");
  SAM6D_REQUIRE(N >= 0 && H > 0 && W > 0, "detections_small_keep: bad sizes");
  if (N == 0) return 0;
  hipLaunchKernelGGL(small_keep_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, boxes, masks, (long)H * W, thr_box,
                     thr_mask, keep);
  SAM6D_LAUNCH_CHECK("detections_small_keep");
}

// idx = nonzero(keep) in increasing order, count[0] = how many (boolean-mask indexing).  One wave, ballot prefix.
__global__ __launch_bounds__(64) void mask_to_indices_kernel(const unsigned char* __restrict__ keep, int N,
                                                             long long* __restrict__ idx, int* __restrict__ count) {
  const int lane = threadIdx.x;
  int base = 0;
  for (int i0 = 0; i0 < N; i0 += 64) {
    const int i = i0 + lane;
    const bool k = i < N && keep[i] != 0;
    const unsigned long long bal = __ballot(k);
    if (k) idx[base + __popcll(bal & ((1ull << lane) - 1ull))] = i;
    base += __popcll(bal);
  }
  if (lane == 0) count[0] = base;
}

extern "C" int sam6d_mask_to_indices(const unsigned char* keep, int N, long long* idx, int* count, void* stream) {
  SAM6D_REQUIRE(keep && idx && count, "mask_to_indices: null pointer");
  SAM6D_REQUIRE(N >= 0, "mask_to_indices: bad size");
  hipLaunchKernelGGL(mask_to_indices_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, keep, N, idx, count);
  SAM6D_LAUNCH_CHECK("mask_to_indices");
}

// out[j, :] = src[idx[j], :] for rows of `row_bytes` bytes of any dtype (Detections.filter on boxes / masks / scores /
// object_ids).  VEC = bytes moved per thread.
template <typename V>
__global__ __launch_bounds__(256) void take_rows_kernel(const V* __restrict__ src, const long long* __restrict__ idx, long per_row,
                                                        long total, long n_src, V* __restrict__ dst) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long j = e / per_row, c = e - j * per_row;
  long a = idx[j];
  if (a < 0) a += n_src;  // python-style negative index
  V z;
  __builtin_memset(&z, 0, sizeof(V));
  dst[e] = (a >= 0 && a < n_src) ? src[a * per_row + c] : z;  // an out-of-range index yields a zero row, never a fault
}

extern "C" int sam6d_take_rows(const void* src, const long long* idx, long n_src, int M, long row_bytes, void* dst,
                               void* stream) {
  SAM6D_REQUIRE(idx && dst && (src || n_src == 0), "take_rows: null pointer");
  SAM6D_REQUIRE(n_src >= 0 && M >= 0 && row_bytes > 0, "take_rows: bad sizes");
  if (M == 0) return 0;
  SAM6D_REQUIRE(n_src > 0, "take_rows: index into an empty tensor");
  hipStream_t s = (hipStream_t)stream;
  const bool a16 = (row_bytes % 16 == 0) && ((((size_t)src) | ((size_t)dst)) & 15) == 0;
  const bool a4 = (row_bytes % 4 == 0) && ((((size_t)src) | ((size_t)dst)) & 3) == 0;
  if (a16) {
    const long per = row_bytes / 16, total = per * M;
    hipLaunchKernelGGL(take_rows_kernel<uint4>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const uint4*)src, idx,
                       per, total, n_src, (uint4*)dst);
  } else if (a4) {
    const long per = row_bytes / 4, total = per * M;
    hipLaunchKernelGGL(take_rows_kernel<unsigned>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const unsigned*)src, idx, per, total, n_src, (unsigned*)dst);
  } else {
    const long total = row_bytes * M;
    hipLaunchKernelGGL(take_rows_kernel<unsigned char>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       (const unsigned char*)src, idx, row_bytes, total, n_src, (unsigned char*)dst);
  }
  SAM6D_LAUNCH_CHECK("take_rows");
}

// ------------------------------------------------------------------------------------------------- NMS
// 1. order: position of box i in (group ascending, score descending, index ascending) -- torch.unique's ascending ids,
//    then torchvision's stable descending sort inside each id.  Rank counting, O(N^2) over L2-resident scores.
__global__ __launch_bounds__(256) void nms_order_kernel(const float* __restrict__ scores, const long long* __restrict__ group,
                                                        int N, int* __restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float si = scores[i];
  const long long gi = group ? group[i] : 0;
  int pos = 0;
  for (int j = 0; j < N; ++j) {
    const float sj = scores[j];
    const long long gj = group ? group[j] : 0;
    pos += (gj < gi) || (gj == gi && (sj > si || (sj == si && j < i)));
  }
  order[pos] = i;
}

// 2. suppression bit matrix over the sorted list: bit c of word (r, cw) = sorted box cw*64+c comes after r, has the
//    same group and IoU > thresh
__global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes, const long long* __restrict__ group,
                                                      const int* __restrict__ order, int N, int nw, float thresh,
                                                      unsigned long long* __restrict__ mask) {
  __shared__ float4 cb[64];
  __shared__ long long cg[64];
  const int rb = blockIdx.y, cw = blockIdx.x, t = threadIdx.x;
  const int r = rb * 64 + t;
  if (cw < rb) {
    if (r < N) mask[(size_t)r * nw + cw] = 0ull;
    return;
  }
  const int cidx = cw * 64 + t;
  if (cidx < N) {
    const int o = order[cidx];
    cb[t] = *reinterpret_cast<const float4*>(boxes + (size_t)o * 4);
    cg[t] = group ? group[o] : 0;
  }
  __syncthreads();
  if (r >= N) return;
  const int o = order[r];
  const float4 a = *reinterpret_cast<const float4*>(boxes + (size_t)o * 4);
  const long long ga = group ? group[o] : 0;
  const float area_a = (a.z - a.x) * (a.w - a.y);
  unsigned long long bits = 0ull;
  const int cn = min(64, N - cw * 64);
  for (int c = (cw == rb ? t + 1 : 0); c < cn; ++c) {
    const float4 b = cb[c];
    if (cg[c] != ga) continue;
    const float w = fmaxf(0.f, fminf(a.z, b.z) - fmaxf(a.x, b.x));
    const float h = fmaxf(0.f, fminf(a.w, b.w) - fmaxf(a.y, b.y));
    const float inter = w * h;
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float ovr = inter / (area_a + area_b - inter);
    if (ovr > thresh) bits |= 1ull << c;
  }
  mask[(size_t)r * nw + cw] = bits;
}

// 3. greedy pass over the sorted list (one workgroup; thread w owns removed-word w)
__global__ __launch_bounds__(1024) void nms_scan_kernel(const unsigned long long* __restrict__ mask, const int* __restrict__ order,
                                                        int N, int nw, long long* __restrict__ keep_idx, int* __restrict__ count) {
  __shared__ unsigned long long removed[1024];
  const int t = threadIdx.x;
  if (t < nw) removed[t] = 0ull;
  __syncthreads();
  int cnt = 0;
  for (int i = 0; i < N; ++i) {
    const bool rem = (removed[i >> 6] >> (i & 63)) & 1ull;  // bit i is final: rows only ever set bits of later boxes
    __syncthreads();
    if (!rem) {
      if (t == 0) keep_idx[cnt] = order[i];
      ++cnt;
      if (t < nw && t >= (i >> 6)) removed[t] |= mask[(size_t)i * nw + t];
    }
    __syncthreads();
  }
  if (t == 0) count[0] = cnt;
}

extern "C" size_t sam6d_nms_workspace_bytes(int N) {
  const size_t nw = (size_t)(N + 63) / 64;
  return (((size_t)N * 4 + 15) & ~(size_t)15) + (size_t)N * nw * 8;
}

extern "C" int sam6d_nms(const float* boxes, const float* scores, const long long* group, int N, float thresh,
                         long long* keep_idx, int* count, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(keep_idx && count && (N == 0 || (boxes && scores && ws)), "nms: null pointer");
  SAM6D_REQUIRE(N >= 0 && N <= 65536, "nms: N must be <= 65536 (got %d)", N);
  SAM6D_REQUIRE(ws_bytes >= sam6d_nms_workspace_bytes(N), "nms: workspace too small");
  SAM6D_REQUIRE(N == 0 || (((size_t)boxes | (size_t)ws) & 15) == 0, "nms: boxes and workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  if (N == 0) {
    const hipError_t e = hipMemsetAsync(count, 0, sizeof(int), s);
    SAM6D_REQUIRE(e == hipSuccess, "nms: memset failed: %s", hipGetErrorString(e));
    return 0;
  }
  const int nw = (N + 63) / 64;
  int* order = (int*)ws;
  unsigned long long* mask = (unsigned long long*)((char*)ws + (((size_t)N * 4 + 15) & ~(size_t)15));
  hipLaunchKernelGGL(nms_order_kernel, dim3((N + 255) / 256), dim3(256), 0, s, scores, group, N, order);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(nw, nw), dim3(64), 0, s, boxes, group, order, N, nw, thresh, mask);
  const int threads = ((nw + 63) / 64) * 64;
  hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(threads), 0, s, mask, order, N, nw, keep_idx, count);
  SAM6D_LAUNCH_CHECK("nms");
}

// =========================================================================================================
// Uncompressed COCO RLE of the detection masks: the `segmentation` field of detection_ism.json
// (mask_to_rle(force_binary_mask(mask)), ISM/model/utils.py:25-43, 199-216; utils/bbox_utils.py:190-192) and its
// inverse as PEM reads it back (cocomask.decode of an uncompressed RLE, PEM/run_inference_custom_pytorch.py:312-317;
// same format as segment_anything/utils/amg.py:107-150).
// Runs are counted over (mask > 0) in COLUMN-major order (i = x*H + y), starting with the zero run (length 0 when
// pixel 0 is set): boundaries p_0 < p_1 < ... are the positions whose value differs from the previous one (value -1 := 0);
// counts = [p_0, p_1 - p_0, ..., H*W - p_last] (a single H*W for an empty mask).
// One workgroup per mask; thread t owns a contiguous chunk of columns = a contiguous piece of the column-major order.
// The reference walks the 307 200 pixels of every mask in a Python loop.
// =========================================================================================================
struct RleChunk {
  int nb;    // boundaries inside the chunk
  int last;  // position of the chunk's last boundary, -1 if none
};

__device__ __forceinline__ RleChunk rle_walk(const float* __restrict__ m, int H, int W, int c0, int c1, int prev_pos,
                                             int* __restrict__ out) {
  RleChunk r{0, -1};
  if (c0 >= c1) return r;
  bool pv = (c0 > 0) ? (m[(size_t)(H - 1) * W + (c0 - 1)] > 0.f) : false;
  for (int c = c0; c < c1; ++c) {
    for (int y = 0; y < H; ++y) {
      const bool v = m[(size_t)y * W + c] > 0.f;
      if (v != pv) {
        const int p = c * H + y;
        if (out) {
          out[r.nb] = p - prev_pos;
          prev_pos = p;
        }
        ++r.nb;
        r.last = p;
        pv = v;
      }
    }
  }
  return r;
}

// exclusive scan of nb and "latest boundary so far" over the 256 chunks (positions grow with the chunk index: max = latest)
__device__ __forceinline__ void rle_scan(RleChunk mine, int t, int* s_nb, int* s_last, int& off, int& prev, int& total, int& last_all) {
  s_nb[t] = mine.nb;
  s_last[t] = mine.last;
  __syncthreads();
  if (t == 0) {
    int acc = 0, lp = -1;
    for (int i = 0; i < 256; ++i) {
      const int n = s_nb[i], l = s_last[i];
      s_nb[i] = acc;
      s_last[i] = lp;
      acc += n;
      lp = l >= 0 ? l : lp;
    }
    s_nb[256] = acc;
    s_last[256] = lp;
  }
  __syncthreads();
  off = s_nb[t];
  prev = s_last[t];
  total = s_nb[256];
  last_all = s_last[256];
}

__global__ __launch_bounds__(256) void rle_count_kernel(const float* __restrict__ masks, int H, int W, int* __restrict__ nruns) {
  __shared__ int s_nb[257], s_last[257];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* m = masks + (size_t)n * H * W;
  const int cpt = (W + 255) / 256;
  const RleChunk mine = rle_walk(m, H, W, min(W, t * cpt), min(W, (t + 1) * cpt), 0, nullptr);
  int off, prev, total, last_all;
  rle_scan(mine, t, s_nb, s_last, off, prev, total, last_all);
  if (t == 0) nruns[n] = total + 1;
}

__global__ __launch_bounds__(256) void rle_encode_kernel(const float* __restrict__ masks, int H, int W,
                                                         const long long* __restrict__ offsets, int* __restrict__ counts) {
  __shared__ int s_nb[257], s_last[257];
  const int n = blockIdx.x, t = threadIdx.x;
  const float* m = masks + (size_t)n * H * W;
  const int cpt = (W + 255) / 256;
  const int c0 = min(W, t * cpt), c1 = min(W, (t + 1) * cpt);
  const RleChunk mine = rle_walk(m, H, W, c0, c1, 0, nullptr);
  int off, prev, total, last_all;
  rle_scan(mine, t, s_nb, s_last, off, prev, total, last_all);
  int* out = counts + offsets[n];
  if ((long long)(total + 1) != offsets[n + 1] - offsets[n]) return;  // offsets do not belong to these masks: write nothing
  rle_walk(m, H, W, c0, c1, prev >= 0 ? prev : 0, out + off);
  if (t == 0) out[total] = H * W - (last_all >= 0 ? last_all : 0);
}

// decode, step 1: ends[o] = inclusive prefix sum of the mask's counts (one workgroup per mask, chunked scan)
__global__ __launch_bounds__(256) void rle_prefix_kernel(const int* __restrict__ counts, const long long* __restrict__ offsets,
                                                         int* __restrict__ ends) {
  __shared__ long long s_sum[257];
  const int n = blockIdx.x, t = threadIdx.x;
  const long long o0 = offsets[n], o1 = offsets[n + 1];
  const long long len = o1 - o0, per = (len + 255) / 256;
  const long long a = min(len, t * per), b = min(len, (t + 1) * per);
  long long s = 0;
  for (long long i = a; i < b; ++i) s += counts[o0 + i];
  s_sum[t] = s;
  __syncthreads();
  if (t == 0) {
    long long acc = 0;
    for (int i = 0; i < 256; ++i) {
      const long long v = s_sum[i];
      s_sum[i] = acc;
      acc += v;
    }
  }
  __syncthreads();
  long long run = s_sum[t];
  for (long long i = a; i < b; ++i) {
    run += counts[o0 + i];
    ends[o0 + i] = (int)min(run, (long long)0x7fffffff);
  }
}

// decode, step 2: pixel (y, x) sits at column-major position i = x*H + y; it belongs to run j = first j with ends[j] > i and
// is set when j is odd.  Positions past the last run (counts that do not add up to H*W) come back 0.
__global__ __launch_bounds__(256) void rle_fill_kernel(const int* __restrict__ ends, const long long* __restrict__ offsets, int H,
                                                       int W, unsigned char* __restrict__ masks) {
  const int n = blockIdx.y;
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long long)H * W) return;
  const int y = (int)(e / W), x = (int)(e % W);
  const int i = x * H + y;
  const long long o0 = offsets[n];
  const int len = (int)(offsets[n + 1] - o0);
  const int* en = ends + o0;
  int lo = 0, hi = len;  // first j in [0, len) with en[j] > i
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (en[mid] > i) hi = mid; else lo = mid + 1;
  }
  masks[(size_t)n * H * W + e] = (lo < len) ? (unsigned char)(lo & 1) : (unsigned char)0;
}

extern "C" int sam6d_mask_rle_count(const float* masks, int N, int H, int W, int* nruns, void* stream) {
  SAM6D_REQUIRE(masks && nruns, "mask_rle_count: null pointer");
  SAM6D_REQUIRE(N >= 0 && H > 0 && W > 0 && (long)H * W < 2147483647L, "mask_rle_count: bad sizes N=%d H=%d W=%d", N, H, W);
  if (N == 0) return 0;
  hipLaunchKernelGGL(rle_count_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, masks, H, W, nruns);
  SAM6D_LAUNCH_CHECK("mask_rle_count");
}

extern "C" int sam6d_mask_rle_encode(const float* masks, int N, int H, int W, const long long* offsets, int* counts, void* stream) {
  SAM6D_REQUIRE(masks && offsets && counts, "mask_rle_encode: null pointer");
  SAM6D_REQUIRE(N >= 0 && H > 0 && W > 0 && (long)H * W < 2147483647L, "mask_rle_encode: bad sizes N=%d H=%d W=%d", N, H, W);
  if (N == 0) return 0;
  hipLaunchKernelGGL(rle_encode_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, masks, H, W, offsets, counts);
  SAM6D_LAUNCH_CHECK("mask_rle_encode");
}

extern "C" int sam6d_mask_rle_decode(const int* counts, const long long* offsets, int N, int H, int W, int* ws_ends,
                                     unsigned char* masks, void* stream) {
  SAM6D_REQUIRE(counts && offsets && ws_ends && masks, "mask_rle_decode: null pointer");
  SAM6D_REQUIRE(N >= 0 && N <= 65535 && H > 0 && W > 0 && (long)H * W < 2147483647L, "mask_rle_decode: bad sizes N=%d H=%d W=%d", N, H, W);
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rle_prefix_kernel, dim3(N), dim3(256), 0, s, counts, offsets, ws_ends);
  hipLaunchKernelGGL(rle_fill_kernel, dim3((unsigned)(((long)H * W + 255) / 256), N), dim3(256), 0, s, ws_ends, offsets, H, W, masks);
  SAM6D_LAUNCH_CHECK("mask_rle_decode");
}
