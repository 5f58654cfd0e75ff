// ISM template scoring for gfx950:
//   PairwiseSimilarity / compute_semantic_score / best_template_pose   ISM/model/loss.py:27-44, ISM/model/detector.py:198-207,260-296
//   compute_straight (appearance) / compute_visible_ratio             ISM/model/loss.py:52-76, detector.py:298-322
//   project_template_to_image / Calculate_the_query_translation       detector.py:209-246, ISM/utils/trimesh_utils.py:77-105
//   compute_iou                                                       ISM/utils/bbox_utils.py:197-222
// All HBM-bound reductions (wave butterflies, coalesced 16-byte lanes); the 256x256x1024 patch-similarity contraction
// runs on the matrix cores through gemm_nt.
#include "common.h"
#include "../../include/sam6d_hip.h"

// ---------------------------------------------------------------------------------------------------------------
// cosine similarity of every query descriptor with every template descriptor, clamped to [0,1].
// The reference L2-normalises both sides and then calls F.cosine_similarity (which divides by the norms again,
// eps 1e-8); both steps are reproduced.  One wave per (query, template) pair.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ism_cosine_kernel(const float* __restrict__ q, const float* __restrict__ ref, int D,
                                                         int NT, long total, float* __restrict__ out) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);  // over Nq * (No*Nt)
  if (w >= total) return;
  const int lane = threadIdx.x & 63;
  const long iq = w / NT;
  const long it = w % NT;
  const float* a = q + iq * D;
  const float* b = ref + it * D;
  float sa = 0.f, sb = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + c);
    const float4 y = *reinterpret_cast<const float4*>(b + c);
    sa += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
    sb += (y.x * y.x + y.y * y.y) + (y.z * y.z + y.w * y.w);
  }
  const float na = fmaxf(sqrtf(wave_sum(sa)), 1e-12f), nb = fmaxf(sqrtf(wave_sum(sb)), 1e-12f);  // F.normalize
  float dot = 0.f, s2a = 0.f, s2b = 0.f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 x = *reinterpret_cast<const float4*>(a + c);
    const float4 y = *reinterpret_cast<const float4*>(b + c);
    const float x0 = x.x / na, x1 = x.y / na, x2 = x.z / na, x3 = x.w / na;
    const float y0 = y.x / nb, y1 = y.y / nb, y2 = y.z / nb, y3 = y.w / nb;
    dot += (x0 * y0 + x1 * y1) + (x2 * y2 + x3 * y3);
    s2a += (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
    s2b += (y0 * y0 + y1 * y1) + (y2 * y2 + y3 * y3);
  }
  dot = wave_sum(dot);
  s2a = wave_sum(s2a);
  s2b = wave_sum(s2b);
  if (lane == 0) {
    const float c = dot / (fmaxf(sqrtf(s2a), 1e-8f) * fmaxf(sqrtf(s2b), 1e-8f));  // F.cosine_similarity
    out[w] = fminf(fmaxf(c, 0.f), 1.f);
  }
}

extern "C" int sam6d_ism_cosine(const float* query, const float* ref, int Nq, int No, int Nt, int D, float* scores,
                                void* stream) {
  SAM6D_REQUIRE(query && ref && scores, "ism_cosine: null pointer");
  SAM6D_REQUIRE(Nq >= 0 && No > 0 && Nt > 0 && D > 0 && (D & 3) == 0, "ism_cosine: bad sizes (D %% 4 == 0)");
  const long total = (long)Nq * No * Nt;
  if (total == 0) return 0;
  hipLaunchKernelGGL(ism_cosine_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, (hipStream_t)stream, query, ref, D,
                     No * Nt, total, scores);
  SAM6D_LAUNCH_CHECK("ism_cosine");
}

// ---------------------------------------------------------------------------------------------------------------
// per query: aggregate template scores per object (avg_5 = mean of the 5 largest / mean / max), argmax object,
// best template of that object (first maximum).  One wave per query; Nt <= 64*4.
// mode: 0 = avg_5, 1 = mean, 2 = max   (detector.py:265-277)
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ism_semantic_kernel(const float* __restrict__ scores, int No, int Nt, int mode,
                                                          float* __restrict__ sem, int* __restrict__ obj, int* __restrict__ best) {
  const int q = blockIdx.x, lane = threadIdx.x;
  float bscore = -INFINITY;
  int bobj = 0, bbest = 0;
  for (int o = 0; o < No; ++o) {
    const float* s = scores + ((size_t)q * No + o) * Nt;
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = (lane + 64 * u < Nt) ? s[lane + 64 * u] : -INFINITY;
    // arg-max template (first maximum)
    float mv = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (v[u] > mv) { mv = v[u]; mi = lane + 64 * u; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
      const float ov = __shfl_xor(mv, d, 64);
      const int oi = __shfl_xor(mi, d, 64);
      if (ov > mv || (ov == mv && oi < mi)) { mv = ov; mi = oi; }
    }
    float agg;
    if (mode == 2) {
      agg = mv;
    } else if (mode == 1) {
      float t = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) t += (lane + 64 * u < Nt) ? v[u] : 0.f;
      agg = wave_sum(t) / (float)Nt;
    } else {  // mean of the 5 largest, summed in descending order like torch.topk(...)[0].mean()
      float t = 0.f;
      const int k = Nt < 5 ? Nt : 5;
      for (int r = 0; r < k; ++r) {
        float lm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        const float gm = wave_max(lm);
        t += gm;
        // remove ONE occurrence of gm (lowest lane holding it)
        const unsigned long long has = __ballot(lm == gm);
        if (lane == (int)(__ffsll((long long)has) - 1)) {
          bool done = false;
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (!done && v[u] == gm) { v[u] = -INFINITY; done = true; }
        }
      }
      agg = t / (float)k;
    }
    if (agg > bscore) { bscore = agg; bobj = o; bbest = mi; }
  }
  if (lane == 0) {
    sem[q] = bscore;
    obj[q] = bobj;
    best[q] = bbest;
  }
}

// ordered compaction of the queries whose score exceeds the confidence threshold (detector.py:284-287); Nq <= 1024.
__global__ __launch_bounds__(1024) void ism_select_kernel(const float* __restrict__ sem, int Nq, float thresh,
                                                          int* __restrict__ sel, int* __restrict__ nsel) {
  __shared__ int wcnt[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const bool keep = (t < Nq) && (sem[t] > thresh);
  const unsigned long long m = __ballot(keep);
  if (lane == 0) wcnt[wave] = __popcll(m);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wcnt[w];
  if (keep) sel[base + __popcll(m & ((1ull << lane) - 1ull))] = t;
  if (t == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += wcnt[w];
    *nsel = tot;
  }
}

// ism_select_kernel with the survivors' values written out compacted, as the int64 / float tensors the caller indexes with
// (detector.py:284-296: idx_selected_proposals, pred_idx_objects, semantic_score, best_template): no follow-up gathers / casts.
__global__ __launch_bounds__(1024) void ism_select_compact_kernel(const float* __restrict__ sem, const int* __restrict__ obj,
                                                                  const int* __restrict__ best, int Nq, float thresh,
                                                                  long long* __restrict__ sel, long long* __restrict__ obj_sel,
                                                                  float* __restrict__ sem_sel, long long* __restrict__ best_sel,
                                                                  int* __restrict__ nsel) {
  __shared__ int wcnt[16];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float v = t < Nq ? sem[t] : 0.f;
  const bool keep = (t < Nq) && (v > thresh);
  const unsigned long long m = __ballot(keep);
  if (lane == 0) wcnt[wave] = __popcll(m);
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wcnt[w];
  if (keep) {
    const int k = base + __popcll(m & ((1ull << lane) - 1ull));
    sel[k] = t;
    obj_sel[k] = obj[t];
    sem_sel[k] = v;
    best_sel[k] = best[t];
  }
  if (t == 0) {
    int tot = 0;
    for (int w = 0; w < 16; ++w) tot += wcnt[w];
    *nsel = tot;
  }
}

extern "C" int sam6d_ism_semantic_compact(const float* scores, int Nq, int No, int Nt, int mode, float thresh, float* sem_ws, int* obj_ws,
                                          int* best_ws, long long* sel, long long* obj_sel, float* sem_sel, long long* best_sel, int* nsel,
                                          void* stream) {
  SAM6D_REQUIRE(scores && sem_ws && obj_ws && best_ws && sel && obj_sel && sem_sel && best_sel && nsel, "ism_semantic_compact: null pointer");
  SAM6D_REQUIRE(Nq >= 0 && Nq <= 1024 && No > 0 && Nt > 0 && Nt <= 256 && mode >= 0 && mode <= 2,
                "ism_semantic_compact: need Nq <= 1024, Nt <= 256, mode in {0 avg_5, 1 mean, 2 max}");
  hipStream_t s = (hipStream_t)stream;
  if (Nq > 0) hipLaunchKernelGGL(ism_semantic_kernel, dim3(Nq), dim3(64), 0, s, scores, No, Nt, mode, sem_ws, obj_ws, best_ws);
  hipLaunchKernelGGL(ism_select_compact_kernel, dim3(1), dim3(1024), 0, s, sem_ws, obj_ws, best_ws, Nq, thresh, sel, obj_sel, sem_sel,
                     best_sel, nsel);
  SAM6D_LAUNCH_CHECK("ism_semantic_compact");
}

extern "C" int sam6d_ism_semantic(const float* scores, int Nq, int No, int Nt, int mode, float thresh, float* sem, int* obj,
                                  int* best, int* sel, int* nsel, void* stream) {
  SAM6D_REQUIRE(scores && sem && obj && best && sel && nsel, "ism_semantic: null pointer");
  SAM6D_REQUIRE(Nq >= 0 && Nq <= 1024 && No > 0 && Nt > 0 && Nt <= 256 && mode >= 0 && mode <= 2,
                "ism_semantic: need Nq <= 1024, Nt <= 256, mode in {0 avg_5, 1 mean, 2 max}");
  hipStream_t s = (hipStream_t)stream;
  if (Nq > 0) hipLaunchKernelGGL(ism_semantic_kernel, dim3(Nq), dim3(64), 0, s, scores, No, Nt, mode, sem, obj, best);
  hipLaunchKernelGGL(ism_select_kernel, dim3(1), dim3(1024), 0, s, sem, Nq, thresh, sel, nsel);
  SAM6D_LAUNCH_CHECK("ism_semantic");
}

// ---------------------------------------------------------------------------------------------------------------
// patch-similarity reductions over sim (Ns, P, P) = q_appe @ ref_appe^T  (rows: query patches, cols: template patches)
//   appearance  = clamp( sum_rows max_cols sim / (count_nonzero(rowsum(q_appe)) + 1e-6), 0, 1 )      loss.py:52-62
//   visible     = count(colmax > thr and colmax != 0) / (count_nonzero(colmax) + 1e-6), colmax = max_rows sim   :64-76
// One workgroup per proposal.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ism_patch_scores_kernel(const float* __restrict__ sim, const float* __restrict__ q_appe,
                                                               int P, int D, float thr, float* __restrict__ appe,
                                                               float* __restrict__ vis) {
  __shared__ float red[3][4];
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* S = sim + (size_t)i * P * P;
  // row maxima (one wave per row, strided) and the query-patch occupancy
  float rsum = 0.f, nz = 0.f;
  for (int r = wave; r < P; r += 4) {
    float m = -INFINITY;
    for (int c = lane; c < P; c += 64) m = fmaxf(m, S[(size_t)r * P + c]);
    m = wave_max(m);
    const float* qr = q_appe + ((size_t)i * P + r) * D;
    float s = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
      const float4 x = *reinterpret_cast<const float4*>(qr + c);
      s += (x.x + x.y) + (x.z + x.w);
    }
    s = wave_sum(s);
    if (lane == 0) {
      rsum += m;
      nz += (s != 0.f) ? 1.f : 0.f;
    }
  }
  // column maxima: thread per column
  float cv = 0.f, cn = 0.f;
  for (int c = t; c < P; c += 256) {
    float m = -INFINITY;
    for (int r = 0; r < P; ++r) m = fmaxf(m, S[(size_t)r * P + c]);
    cn += (m != 0.f) ? 1.f : 0.f;
    cv += (m > thr && m != 0.f) ? 1.f : 0.f;
  }
  cv = wave_sum(cv);
  cn = wave_sum(cn);
  if (lane == 0) {
    red[0][wave] = rsum;
    red[1][wave] = nz;
    red[2][wave] = cv;
  }
  __shared__ float red2[4];
  if (lane == 0) red2[wave] = cn;
  __syncthreads();
  if (t == 0) {
    const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float n = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float v = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const float vn = (red2[0] + red2[1]) + (red2[2] + red2[3]);
    appe[i] = fminf(fmaxf(a / (n + 1e-6f), 0.f), 1.f);
    vis[i] = v / (vn + 1e-6f);
  }
}

extern "C" int sam6d_ism_patch_scores(const float* sim, const float* q_appe, int Ns, int P, int D, float thred, float* appe,
                                      float* vis, void* stream) {
  SAM6D_REQUIRE(sim && q_appe && appe && vis && Ns >= 0 && P > 0 && D > 0 && (D & 3) == 0, "ism_patch_scores: bad arguments");
  if (Ns == 0) return 0;
  hipLaunchKernelGGL(ism_patch_scores_kernel, dim3(Ns), dim3(256), 0, (hipStream_t)stream, sim, q_appe, P, D, thred, appe, vis);
  SAM6D_LAUNCH_CHECK("ism_patch_scores");
}

// ---------------------------------------------------------------------------------------------------------------
// The same two scores WITHOUT the similarity tensor and without gathering the chosen templates' descriptors ("fuse"):
// proposal p multiplies its P x D query patches with the P x D patches of template (obj[p], best[p]) read IN PLACE from ref_data
// (detector.py:298-308, 310-322 gather them into a new tensor first), on v_mfma_f32_32x32x16_f16 with the fp16 x3 split done while
// the tiles are staged (descriptors are L2-normalised, |x| <= 1: x 2^10, split, products exact in the fp32 accumulator).  The 128 x 128
// tile never leaves the registers: the epilogue reduces it to row maxima over its 64-column wave slice and column maxima over its
// 64-row slice (4 partials per row / column for P = 256) plus the query-patch occupancy flags; ism_patch_finish_kernel merges them in
// the summation order of ism_patch_scores_kernel, so both paths return the same bits given the same products.
// (N, 256, 256) floats = 39 MB written + read and 2 x 157 MB of gathered descriptors per pass disappear.
// ---------------------------------------------------------------------------------------------------------------
typedef float ip_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 ip_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 ip_half4 __attribute__((ext_vector_type(4)));
typedef unsigned ip_u2 __attribute__((ext_vector_type(2)));
#define IP_BK 32
#define IP_LD 40
#define IP_SCALE 1024.0f
__device__ __forceinline__ void ip_split4(const float4 v, ip_half4& hi, ip_half4& lo) {
  unsigned h0, h1, l0, l1;
  sam6d_split2_f16(v.x * IP_SCALE, v.y * IP_SCALE, h0, l0);
  sam6d_split2_f16(v.z * IP_SCALE, v.w * IP_SCALE, h1, l1);
  hi = __builtin_bit_cast(ip_half4, ip_u2{h0, h1});
  lo = __builtin_bit_cast(ip_half4, ip_u2{l0, l1});
}

__global__ __launch_bounds__(256) void ism_patch_fused_kernel(const float* __restrict__ q, const long long* __restrict__ qsel,
                                                              const float* __restrict__ ref, const long long* __restrict__ obj,
                                                              const long long* __restrict__ best, int Nt, int P, int D,
                                                              float* __restrict__ rowpart, float* __restrict__ colpart,
                                                              float* __restrict__ nzflag) {
  __shared__ __attribute__((aligned(16))) _Float16 smem[4 * 128 * IP_LD];
  _Float16* Ah = smem;
  _Float16* Al = Ah + 128 * IP_LD;
  _Float16* Bh = Al + 128 * IP_LD;
  _Float16* Bl = Bh + 128 * IP_LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int p = blockIdx.y, nt = P / 128, tm = blockIdx.x % nt, tn = blockIdx.x / nt, slots = 2 * nt;
  const float* A = q + ((size_t)(qsel ? qsel[p] : p) * P + 128 * tm) * D;
  const float* W = ref + (((size_t)obj[p] * Nt + (size_t)best[p]) * P + 128 * tn) * D;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  ip_f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int sr = t >> 3, sk = (t & 7) * 4;
  float4 va[4], vb[4];
  float rs[4] = {0.f, 0.f, 0.f, 0.f};
  auto fetch = [&](int k0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      va[u] = *reinterpret_cast<const float4*>(A + (size_t)(sr + 32 * u) * D + k0 + sk);
      vb[u] = *reinterpret_cast<const float4*>(W + (size_t)(sr + 32 * u) * D + k0 + sk);
    }
  };
  const int fr = lane & 31, fk = lane >> 5;
  fetch(0);
  for (int k0 = 0; k0 < D; k0 += IP_BK) {
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ip_half4 hi, lo;
      ip_split4(va[u], hi, lo);
      *reinterpret_cast<ip_half4*>(&Ah[(sr + 32 * u) * IP_LD + sk]) = hi;
      *reinterpret_cast<ip_half4*>(&Al[(sr + 32 * u) * IP_LD + sk]) = lo;
      rs[u] += (va[u].x + va[u].y) + (va[u].z + va[u].w);
      ip_split4(vb[u], hi, lo);
      *reinterpret_cast<ip_half4*>(&Bh[(sr + 32 * u) * IP_LD + sk]) = hi;
      *reinterpret_cast<ip_half4*>(&Bl[(sr + 32 * u) * IP_LD + sk]) = lo;
    }
    __syncthreads();
    if (k0 + IP_BK < D) fetch(k0 + IP_BK);
#pragma unroll
    for (int ks = 0; ks < IP_BK; ks += 16) {
      ip_half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const ip_half8*>(&Ah[(wm + 32 * i + fr) * IP_LD + ks + 8 * fk]);
        al[i] = *reinterpret_cast<const ip_half8*>(&Al[(wm + 32 * i + fr) * IP_LD + ks + 8 * fk]);
        bh[i] = *reinterpret_cast<const ip_half8*>(&Bh[(wn + 32 * i + fr) * IP_LD + ks + 8 * fk]);
        bl[i] = *reinterpret_cast<const ip_half8*>(&Bl[(wn + 32 * i + fr) * IP_LD + ks + 8 * fk]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  const float un = 1.0f / (IP_SCALE * IP_SCALE);
  // lane (fr, fk) holds rows (r & 3) + 8 (r >> 2) + 4 fk of column fr of each 32 x 32 tile
  // ---- column maxima over this wave's 64 rows
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) m = fmaxf(m, acc[i][j][r]);
    m = fmaxf(m, xor32_f32(m));
    if (fk == 0) colpart[((size_t)p * P + 128 * tn + wn + 32 * j + fr) * slots + 2 * tm + (wave >> 1)] = m * un;
  }
  // ---- row maxima over this wave's 64 columns
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float m = fmaxf(acc[i][0][r], acc[i][1][r]);
      m = row16_max_dpp(m);
      m = fmaxf(m, xor16_f32(m));
      if (fr == 0) rowpart[((size_t)p * P + 128 * tm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk) * slots + 2 * tn + (wave & 1)] = m * un;
    }
  // ---- occupancy of the query patches (count_nonzero(query.sum(-1)), loss.py:58): the workgroups of the first column tile
  if (tn == 0) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      float s_ = rs[u];
      s_ += __shfl_xor(s_, 1, 64);
      s_ += __shfl_xor(s_, 2, 64);
      s_ += __shfl_xor(s_, 4, 64);
      if ((t & 7) == 0) nzflag[(size_t)p * P + 128 * tm + sr + 32 * u] = (s_ != 0.f) ? 1.f : 0.f;
    }
  }
}

// merge: the summation order of ism_patch_scores_kernel (rows r = w, w + 4, ... per wave w; thread per column, wave sums)
__global__ __launch_bounds__(256) void ism_patch_finish_kernel(const float* __restrict__ rowpart, const float* __restrict__ colpart,
                                                               const float* __restrict__ nzflag, int P, int slots, float thr,
                                                               float* __restrict__ appe, float* __restrict__ vis) {
  __shared__ float red[3][4];
  __shared__ float red2[4];
  const int i = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  float rsum = 0.f, nz = 0.f;
  if (lane == 0) {
    for (int r = wave; r < P; r += 4) {
      float m = -INFINITY;
      for (int s_ = 0; s_ < slots; ++s_) m = fmaxf(m, rowpart[((size_t)i * P + r) * slots + s_]);
      rsum += m;
      nz += nzflag[(size_t)i * P + r];
    }
  }
  float cv = 0.f, cn = 0.f;
  for (int c = t; c < P; c += 256) {
    float m = -INFINITY;
    for (int s_ = 0; s_ < slots; ++s_) m = fmaxf(m, colpart[((size_t)i * P + c) * slots + s_]);
    cn += (m != 0.f) ? 1.f : 0.f;
    cv += (m > thr && m != 0.f) ? 1.f : 0.f;
  }
  cv = wave_sum(cv);
  cn = wave_sum(cn);
  if (lane == 0) {
    red[0][wave] = rsum;
    red[1][wave] = nz;
    red[2][wave] = cv;
    red2[wave] = cn;
  }
  __syncthreads();
  if (t == 0) {
    const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float n = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float v = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    const float vn = (red2[0] + red2[1]) + (red2[2] + red2[3]);
    if (appe) appe[i] = fminf(fmaxf(a / (n + 1e-6f), 0.f), 1.f);
    if (vis) vis[i] = v / (vn + 1e-6f);
  }
}

extern "C" size_t sam6d_ism_patch_fused_workspace_bytes(int Ns, int P) { return (size_t)Ns * P * (2 * (P / 128) * 2 + 1) * 4; }

extern "C" int sam6d_ism_patch_fused(const float* q, const long long* qsel, const float* ref, const long long* obj, const long long* best,
                                     int Ns, int Nt, int P, int D, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(q && ref && obj && best && ws && Ns >= 0 && Nt > 0, "ism_patch_fused: bad arguments");
  SAM6D_REQUIRE(P > 0 && (P % 128) == 0 && D > 0 && (D % IP_BK) == 0, "ism_patch_fused: P must be a multiple of 128, D of 32");
  SAM6D_REQUIRE(((((size_t)q) | ((size_t)ref) | ((size_t)ws)) & 15) == 0 && ws_bytes >= sam6d_ism_patch_fused_workspace_bytes(Ns, P),
                "ism_patch_fused: 16-byte alignment / workspace size (sam6d_ism_patch_fused_workspace_bytes)");
  if (Ns == 0) return 0;
  const int nt = P / 128, slots = 2 * nt;
  float* rowpart = (float*)ws;
  float* colpart = rowpart + (size_t)Ns * P * slots;
  float* nzflag = colpart + (size_t)Ns * P * slots;
  hipLaunchKernelGGL(ism_patch_fused_kernel, dim3(nt * nt, Ns), dim3(256), 0, (hipStream_t)stream, q, qsel, ref, obj, best, Nt, P, D, rowpart,
                     colpart, nzflag);
  SAM6D_LAUNCH_CHECK("ism_patch_fused");
}

extern "C" int sam6d_ism_patch_fused_scores(const void* ws, int Ns, int P, float thred, float* appe, float* vis, void* stream) {
  SAM6D_REQUIRE(ws && Ns >= 0 && P > 0 && (P % 128) == 0 && (appe || vis), "ism_patch_fused_scores: bad arguments");
  if (Ns == 0) return 0;
  const int slots = 2 * (P / 128);
  const float* rowpart = (const float*)ws;
  const float* colpart = rowpart + (size_t)Ns * P * slots;
  const float* nzflag = colpart + (size_t)Ns * P * slots;
  hipLaunchKernelGGL(ism_patch_finish_kernel, dim3(Ns), dim3(256), 0, (hipStream_t)stream, rowpart, colpart, nzflag, P, slots, thred, appe,
                     vis);
  SAM6D_LAUNCH_CHECK("ism_patch_fused_scores");
}

// ---------------------------------------------------------------------------------------------------------------
// Query translation = mean back-projected masked depth (detector.py:234-246, trimesh_utils.py:77-105), then template
// point cloud -> image (detector.py:209-232).  The reference's caller hands K and depth_scale over as float64
// (ISM/run_inference_custom.py:87-94), which makes its whole translation computation float64 until the final
// .to(float32) (detector.py:246); the kernel does the same: per-pixel terms and sums in double, K in double.
// The projection uses K cast to float32, as detector.py:225 does.
// ---------------------------------------------------------------------------------------------------------------
#define ISM_TCH 64  // pixel chunks per proposal
__global__ __launch_bounds__(256) void ism_translate_partial_kernel(const float* __restrict__ masks, const int* __restrict__ depth,
                                                                    const double* __restrict__ K, double scale, int H, int Wd,
                                                                    double* __restrict__ part) {
  __shared__ double red[4][4];
  const int i = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
  const long npx = (long)H * Wd;
  const long per = (npx + ISM_TCH - 1) / ISM_TCH;
  const long p0 = ch * per, p1 = min(npx, p0 + per);
  const float* m = masks + (size_t)i * npx;
  const double cx = K[2], fx = K[0], cy = K[5], fy = K[4];
  double sx = 0, sy = 0, sz = 0, sn = 0;
  for (long p = p0 + t; p < p1; p += 256) {
    const double md = (double)(m[p] * (float)depth[p]);  // mask * depth (exact: 0/1 times an integer < 2^24)
    const double Z = md * scale / 1000.0;
    if (Z > 0.0) {
      const int u = (int)(p % Wd), v = (int)(p / Wd);
      sx += ((double)u - cx) * Z / fx;
      sy += ((double)v - cy) * Z / fy;
      sz += Z;
      sn += 1.0;
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); sn = wave_sum(sn);
  if ((t & 63) == 0) { red[0][t >> 6] = sx; red[1][t >> 6] = sy; red[2][t >> 6] = sz; red[3][t >> 6] = sn; }
  __syncthreads();
  if (t < 4) part[((size_t)i * ISM_TCH + ch) * 4 + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
}

// one workgroup per proposal: finish the translation, pose the cloud, project, truncate, clamp, bounding box
__global__ __launch_bounds__(256) void ism_project_kernel(const double* __restrict__ part, const float* __restrict__ poses,
                                                          const float* __restrict__ pc, const int* __restrict__ best,
                                                          const int* __restrict__ obj, const double* __restrict__ Kd, int Npc, int H,
                                                          int Wd, int* __restrict__ vu, int* __restrict__ xyxy,
                                                          float* __restrict__ translate, int nch = ISM_TCH, double zscale = 0.0) {
  __shared__ float tr[3];
  float K[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) K[e] = (float)Kd[e];
  __shared__ int bb[4][4];
  const int i = blockIdx.x, t = threadIdx.x;
  if (t < 3) {
    if (zscale == 0.0) {  // partials = sums of the per-pixel coordinates (ism_translate_partial_kernel)
      double s = 0, n = 0;
      for (int c = 0; c < nch; ++c) {
        s += part[((size_t)i * nch + c) * 4 + t];
        n += part[((size_t)i * nch + c) * 4 + 3];
      }
      tr[t] = (float)(s / (n + 1e-8));
    } else {
      // partials = (sum u md, sum v md, sum md, count) of the masked depth md (ism_translate_fast_kernel; exact: integers below 2^53):
      // sum_p (u - cx) Z / fx with Z = md zscale  =  (sum u md - cx sum md) zscale / fx
      double su = 0, sv = 0, sm = 0, n = 0;
      for (int c = 0; c < nch; ++c) {
        const double* q = part + ((size_t)i * nch + c) * 4;
        su += q[0]; sv += q[1]; sm += q[2]; n += q[3];
      }
      const double v = t == 0 ? (su - Kd[2] * sm) * zscale / Kd[0] : t == 1 ? (sv - Kd[5] * sm) * zscale / Kd[4] : sm * zscale;
      tr[t] = (float)(v / (n + 1e-8));
    }
    translate[i * 3 + t] = tr[t];
  }
  __syncthreads();
  const float* R = poses + (size_t)best[i] * 16;  // (4,4) row-major, rotation = [0:3,0:3]
  const float* P = pc + (size_t)obj[i] * Npc * 3;
  int x0 = 0x7fffffff, y0 = 0x7fffffff, x1 = -0x7fffffff, y1 = -0x7fffffff;
  for (int k = t; k < Npc; k += 256) {
    const float a = P[k * 3], b = P[k * 3 + 1], c = P[k * 3 + 2];
    // posed = R @ p (torch matmul K=3 chain) + translate
    const float X = fmaf(R[2], c, fmaf(R[1], b, R[0] * a)) + tr[0];
    const float Y = fmaf(R[6], c, fmaf(R[5], b, R[4] * a)) + tr[1];
    const float Z = fmaf(R[10], c, fmaf(R[9], b, R[8] * a)) + tr[2];
    // homogeneous image point K @ posed, divided by its last component
    const float hx = fmaf(K[2], Z, fmaf(K[1], Y, K[0] * X));
    const float hy = fmaf(K[5], Z, fmaf(K[4], Y, K[3] * X));
    const float hz = fmaf(K[8], Z, fmaf(K[7], Y, K[6] * X));
    int px = (int)(hx / hz), py = (int)(hy / hz);  // .to(torch.int): truncation toward zero
    px = min(max(px, 0), Wd - 1);
    py = min(max(py, 0), H - 1);
    vu[((size_t)i * Npc + k) * 2] = px;
    vu[((size_t)i * Npc + k) * 2 + 1] = py;
    x0 = min(x0, px); y0 = min(y0, py); x1 = max(x1, px); y1 = max(y1, py);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    x0 = min(x0, __shfl_xor(x0, d, 64)); y0 = min(y0, __shfl_xor(y0, d, 64));
    x1 = max(x1, __shfl_xor(x1, d, 64)); y1 = max(y1, __shfl_xor(y1, d, 64));
  }
  if ((t & 63) == 0) { bb[0][t >> 6] = x0; bb[1][t >> 6] = y0; bb[2][t >> 6] = x1; bb[3][t >> 6] = y1; }
  __syncthreads();
  if (t == 0) {
    xyxy[i * 4 + 0] = min(min(bb[0][0], bb[0][1]), min(bb[0][2], bb[0][3]));
    xyxy[i * 4 + 1] = min(min(bb[1][0], bb[1][1]), min(bb[1][2], bb[1][3]));
    xyxy[i * 4 + 2] = max(max(bb[2][0], bb[2][1]), max(bb[2][2], bb[2][3]));
    xyxy[i * 4 + 3] = max(max(bb[3][0], bb[3][1]), max(bb[3][2], bb[3][3]));
  }
}

// Round 4: the masked-depth sums as a 16-byte-per-lane stream.  The old kernel read 4-byte mask floats and did three fp64 divisions per
// pixel (0.2 ms for 246 MB: 1.2 TB/s).  Here a workgroup owns a chunk of 4096 pixels (16 per lane) of ISM_PG consecutive proposals: the
// depth chunk is loaded once into registers, every proposal costs one 16-byte mask load per lane (uint8 masks: 16 pixels; fp32 masks:
// four float4 loads), and the per-pixel work is a select, one fp64 add and one fp64 fma -- the divisions move to the finish
// (sum_p (u - cx) Z_p / fx = (sum u md - cx sum md) zscale / fx with exact integer sums).  Masks are read in place through an
// optional index (the proposals selected by the class-token filter: no gathered (Ns,H,W) copy).  Needs W % 16 == 0 (a lane's 16
// pixels share an image row) and 16-byte aligned rows; depth_scale > 0.
#define ISM_PG 8
template <typename MT>
__global__ __launch_bounds__(256) void ism_translate_fast_kernel(const MT* __restrict__ masks, const long long* __restrict__ midx,
                                                                 const int* __restrict__ depth, int Ns, long npx, int Wd, int nch,
                                                                 double* __restrict__ part) {
  __shared__ double red[4][4];
  const int ch = blockIdx.x, g0 = blockIdx.y * ISM_PG, t = threadIdx.x;
  const long p0 = (long)ch * 4096 + (long)t * 16;
  const bool inb = p0 < npx;  // (npx % 16 == 0: a lane's 16 pixels are all inside or all outside)
  float dz[16];
  {
    const uint4* dp = reinterpret_cast<const uint4*>(depth + (inb ? p0 : 0));
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint4 v = dp[q];
      dz[4 * q] = (float)(int)v.x; dz[4 * q + 1] = (float)(int)v.y; dz[4 * q + 2] = (float)(int)v.z; dz[4 * q + 3] = (float)(int)v.w;
    }
  }
  const int u0 = (int)(p0 % Wd), v0 = (int)(p0 / Wd);
  for (int gi = 0; gi < ISM_PG; ++gi) {
    const int i = g0 + gi;
    if (i >= Ns) break;  // (block-uniform)
    const MT* m = masks + (size_t)(midx ? midx[i] : i) * npx + (inb ? p0 : 0);
    float mk[16];
    if constexpr (sizeof(MT) == 1) {
      typedef unsigned ism_u4 __attribute__((ext_vector_type(4)));
      const ism_u4 w = __builtin_nontemporal_load(reinterpret_cast<const ism_u4*>(m));
      const unsigned ww[4] = {w[0], w[1], w[2], w[3]};
#pragma unroll
      for (int j = 0; j < 16; ++j) mk[j] = ((ww[j >> 2] >> (8 * (j & 3))) & 0xffu) ? 1.0f : 0.0f;
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        typedef float ism_f4 __attribute__((ext_vector_type(4)));
        const ism_f4 w = __builtin_nontemporal_load(reinterpret_cast<const ism_f4*>(m) + q);
        mk[4 * q] = w[0]; mk[4 * q + 1] = w[1]; mk[4 * q + 2] = w[2]; mk[4 * q + 3] = w[3];
      }
    }
    double sm = 0.0, sj = 0.0;
    float cnt = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float mdf = mk[j] * dz[j];  // mask * depth as the reference forms it (fp32 product; exact for 0/1 masks, depth < 2^24)
      const bool ok = inb && mdf > 0.f; // Z = md * zscale > 0 with zscale > 0
      const double md = ok ? (double)mdf : 0.0;
      sm += md;
      sj = fma((double)j, md, sj);
      cnt += ok ? 1.f : 0.f;
    }
    double su = fma((double)u0, sm, sj), sv = (double)v0 * sm, sn = (double)cnt;
    su = wave_sum(su); sv = wave_sum(sv); sm = wave_sum(sm); sn = wave_sum(sn);
    if (gi) __syncthreads();
    if ((t & 63) == 0) { red[0][t >> 6] = su; red[1][t >> 6] = sv; red[2][t >> 6] = sm; red[3][t >> 6] = sn; }
    __syncthreads();
    if (t < 4) part[((size_t)i * nch + ch) * 4 + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
  }
}

extern "C" size_t sam6d_ism_project_workspace_doubles(int Ns, int H, int W) {
  const long npx = (long)H * W;
  const long nch = (npx + 4095) / 4096;
  return (size_t)Ns * (size_t)(nch > ISM_TCH ? nch : ISM_TCH) * 4;
}

extern "C" int sam6d_ism_project2(const void* masks, int mask_bytes, const long long* mask_index, const int* depth, const double* K,
                                  double depth_scale, const float* poses, const float* pointcloud, const int* best, const int* obj,
                                  int Ns, int H, int W, int Npc, double* part_ws, int* image_vu, int* xyxy, float* translate,
                                  void* stream) {
  SAM6D_REQUIRE(masks && depth && K && poses && pointcloud && best && obj && part_ws && image_vu && xyxy && translate,
                "ism_project2: null pointer");
  SAM6D_REQUIRE(mask_bytes == 1 || mask_bytes == 4, "ism_project2: masks must be uint8 / bool (1 byte) or float32 (4 bytes) per pixel");
  SAM6D_REQUIRE(Ns >= 0 && Ns <= 65535 * ISM_PG && H > 0 && W > 0 && Npc > 0, "ism_project2: bad sizes");
  if (Ns == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const long npx = (long)H * W;
  const bool fast = (W % 16) == 0 && depth_scale > 0.0 && ((((size_t)masks) | ((size_t)depth)) & 15) == 0;
  if (!fast) {
    SAM6D_REQUIRE(mask_bytes == 4 && !mask_index,
                  "ism_project2: the 16-byte path needs W %% 16 == 0, depth_scale > 0 and 16-byte aligned buffers; the general path "
                  "(sam6d_ism_project) takes float32 masks without an index");
    return sam6d_ism_project((const float*)masks, depth, K, depth_scale, poses, pointcloud, best, obj, Ns, H, W, Npc, part_ws, image_vu,
                             xyxy, translate, stream);
  }
  const int nch = (int)((npx + 4095) / 4096);
  const dim3 grid(nch, (Ns + ISM_PG - 1) / ISM_PG);
  if (mask_bytes == 1)
    hipLaunchKernelGGL(ism_translate_fast_kernel<unsigned char>, grid, dim3(256), 0, s, (const unsigned char*)masks, mask_index, depth, Ns,
                       npx, W, nch, part_ws);
  else
    hipLaunchKernelGGL(ism_translate_fast_kernel<float>, grid, dim3(256), 0, s, (const float*)masks, mask_index, depth, Ns, npx, W, nch,
                       part_ws);
  hipLaunchKernelGGL(ism_project_kernel, dim3(Ns), dim3(256), 0, s, part_ws, poses, pointcloud, best, obj, K, Npc, H, W, image_vu, xyxy,
                     translate, nch, depth_scale / 1000.0);
  SAM6D_LAUNCH_CHECK("ism_project2");
}

extern "C" int sam6d_ism_project(const float* masks, const int* depth, const double* K, double depth_scale, const float* poses,
                                 const float* pointcloud, const int* best, const int* obj, int Ns, int H, int W, int Npc,
                                 double* part_ws, int* image_vu, int* xyxy, float* translate, void* stream) {
  SAM6D_REQUIRE(masks && depth && K && poses && pointcloud && best && obj && part_ws && image_vu && xyxy && translate,
                "ism_project: null pointer");
  SAM6D_REQUIRE(Ns >= 0 && Ns <= 65535 && H > 0 && W > 0 && Npc > 0, "ism_project: bad sizes");
  if (Ns == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ism_translate_partial_kernel, dim3(ISM_TCH, Ns), dim3(256), 0, s, masks, depth, K, depth_scale, H, W, part_ws);
  hipLaunchKernelGGL(ism_project_kernel, dim3(Ns), dim3(256), 0, s, part_ws, poses, pointcloud, best, obj, K, Npc, H, W, image_vu,
                     xyxy, translate);
  SAM6D_LAUNCH_CHECK("ism_project");
}

// depth_image_to_pointcloud_translate_torch as the reference defines it (trimesh_utils.py:77-105): N already-masked depth maps in, N mean
// back-projected points out -- one launch for all maps (the per-pixel terms and sums in double like the kernel above).
__global__ __launch_bounds__(256) void ism_translate_maps_kernel(const float* __restrict__ md, const double* __restrict__ K, double scale,
                                                                 int H, int Wd, double* __restrict__ part) {
  __shared__ double red[4][4];
  const int i = blockIdx.y, ch = blockIdx.x, t = threadIdx.x;
  const long npx = (long)H * Wd;
  const long per = (npx + ISM_TCH - 1) / ISM_TCH;
  const long p0 = ch * per, p1 = min(npx, p0 + per);
  const float* m = md + (size_t)i * npx;
  const double cx = K[2], fx = K[0], cy = K[5], fy = K[4];
  double sx = 0, sy = 0, sz = 0, sn = 0;
  for (long p = p0 + t; p < p1; p += 256) {
    const double Z = (double)m[p] * scale / 1000.0;
    if (Z > 0.0) {
      const int u = (int)(p % Wd), v = (int)(p / Wd);
      sx += ((double)u - cx) * Z / fx;
      sy += ((double)v - cy) * Z / fy;
      sz += Z;
      sn += 1.0;
    }
  }
  sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz); sn = wave_sum(sn);
  if ((t & 63) == 0) { red[0][t >> 6] = sx; red[1][t >> 6] = sy; red[2][t >> 6] = sz; red[3][t >> 6] = sn; }
  __syncthreads();
  if (t < 4) part[((size_t)i * ISM_TCH + ch) * 4 + t] = (red[t][0] + red[t][1]) + (red[t][2] + red[t][3]);
}
__global__ __launch_bounds__(64) void ism_translate_finish_kernel(const double* __restrict__ part, int N, float* __restrict__ translate) {
  const int e = blockIdx.x * 64 + threadIdx.x;  // over N * 3
  if (e >= N * 3) return;
  const int i = e / 3, t = e % 3;
  double s = 0, n = 0;
  for (int c = 0; c < ISM_TCH; ++c) {
    s += part[((size_t)i * ISM_TCH + c) * 4 + t];
    n += part[((size_t)i * ISM_TCH + c) * 4 + 3];
  }
  translate[e] = (float)(s / (n + 1e-8));
}

extern "C" int sam6d_ism_translate_maps(const float* masked_depth, const double* K, double depth_scale, int N, int H, int W,
                                        double* part_ws, float* translate, void* stream) {
  SAM6D_REQUIRE(masked_depth && K && part_ws && translate, "ism_translate_maps: null pointer");
  SAM6D_REQUIRE(N >= 0 && N <= 65535 && H > 0 && W > 0, "ism_translate_maps: bad sizes");
  if (N == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ism_translate_maps_kernel, dim3(ISM_TCH, N), dim3(256), 0, s, masked_depth, K, depth_scale, H, W, part_ws);
  hipLaunchKernelGGL(ism_translate_finish_kernel, dim3(cdiv(N * 3, 64)), dim3(64), 0, s, part_ws, N, translate);
  SAM6D_LAUNCH_CHECK("ism_translate_maps");
}

// IoU of the projected-template box with the proposal box (bbox_utils.py:197-222), integer arithmetic as in the
// reference; `all_positive` (device int, pre-set to 1 here) is cleared when any pair has a non-positive overlap --
// the caller turns that into the reference's scalar-0.0 result (:214-220).  final = (sem + appe + iou*vis)/(2 + vis).
__global__ void ism_iou_kernel(const int* __restrict__ a, const long long* __restrict__ b, int Ns, float* __restrict__ iou,
                               int* __restrict__ all_positive) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ns) return;
  const long long ax0 = a[i * 4], ay0 = a[i * 4 + 1], ax1 = a[i * 4 + 2], ay1 = a[i * 4 + 3];
  const long long bx0 = b[i * 4], by0 = b[i * 4 + 1], bx1 = b[i * 4 + 2], by1 = b[i * 4 + 3];
  const long long w = min(ax1, bx1) - max(ax0, bx0), h = min(ay1, by1) - max(ay0, by0);
  if (!(w > 0 && h > 0)) atomicAnd(all_positive, 0);
  const long long inter = w * h, aa = (ax1 - ax0) * (ay1 - ay0), ab = (bx1 - bx0) * (by1 - by0);
  iou[i] = (float)inter / (float)(aa + ab - inter);
}

__global__ void set_int_kernel(int* p, int v) { *p = v; }

extern "C" int sam6d_ism_iou(const int* xyxy, const long long* boxes, int Ns, float* iou, int* all_positive, void* stream) {
  SAM6D_REQUIRE(xyxy && boxes && iou && all_positive && Ns >= 0, "ism_iou: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s, all_positive, 1);
  if (Ns > 0) hipLaunchKernelGGL(ism_iou_kernel, dim3(cdiv(Ns, 256)), dim3(256), 0, s, xyxy, boxes, Ns, iou, all_positive);
  SAM6D_LAUNCH_CHECK("ism_iou");
}

__global__ void ism_final_kernel(const float* __restrict__ sem, const float* __restrict__ appe, const float* __restrict__ geo,
                                 const float* __restrict__ vis, const int* __restrict__ sel, int Ns, float* __restrict__ out,
                                 const int* __restrict__ all_positive) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Ns) return;
  // geo == NULL, or the device flag of sam6d_ism_iou cleared: the scalar-0.0 IoU quirk
  const float g = (geo && (!all_positive || *all_positive)) ? geo[i] : 0.f;
  const float s = sem[sel ? sel[i] : i];
  out[i] = ((s + appe[i]) + g * vis[i]) / ((1.f + 1.f) + vis[i]);
}

extern "C" int sam6d_ism_final_score(const float* sem, const float* appe, const float* geo, const float* vis, const int* sel,
                                     int Ns, float* out, void* stream) {
  SAM6D_REQUIRE(sem && appe && vis && out && Ns >= 0, "ism_final_score: bad arguments");
  if (Ns == 0) return 0;
  hipLaunchKernelGGL(ism_final_kernel, dim3(cdiv(Ns, 256)), dim3(256), 0, (hipStream_t)stream, sem, appe, geo, vis, sel, Ns, out, nullptr);
  SAM6D_LAUNCH_CHECK("ism_final_score");
}

extern "C" int sam6d_ism_final_score_flag(const float* sem, const float* appe, const float* geo, const float* vis, const int* all_positive,
                                          int Ns, float* out, void* stream) {
  SAM6D_REQUIRE(sem && appe && geo && vis && all_positive && out && Ns >= 0, "ism_final_score_flag: bad arguments");
  if (Ns == 0) return 0;
  hipLaunchKernelGGL(ism_final_kernel, dim3(cdiv(Ns, 256)), dim3(256), 0, (hipStream_t)stream, sem, appe, geo, vis, nullptr, Ns, out,
                     all_positive);
  SAM6D_LAUNCH_CHECK("ism_final_score_flag");
}
