// Fine-stage feature similarity + soft assignment as one pipeline (PEM/utils/model_utils.py:131-153 compute_feature_similarity,
// :308-331 the head of compute_fine_Rt), for n = 2049 tokens per cloud (bg token + 2048 dense points), 256 channels.
//
//   att[n][m] = normalize(f1[n]) . normalize(f2[m]) / temp;  S = softmax(att, 2) * softmax(att, 1);
//   l1 = argmax_m S[n >= 1, :],  l2 = argmax_n S[:, m >= 1];  A = S[1:,1:] [l1 > 0] [l2 > 0];  weight = sum_m A;
//   pred = (A / (weight + 1e-6)) pts2
//
// Before: the 32 x 2049 x 2049 matrix (537 MB) was written by the generic GEMM (4-byte stores: rows of 2049 floats are never
// 16-byte aligned) and read five times (row stats, column stats, row labels, column labels, assignment).  Here it is written ONCE and
// read TWICE:
//   * |att| <= 1 / temp because both sides are L2-normalised, so softmax needs no running maximum: with the fixed shift c = 1 / temp,
//     E = exp(att - c) is in [e^-2c, 1] and softmax(att, 2)[n][m] = E[n][m] / sum_m E[n][m] (same for columns).  The similarity GEMM
//     stores E and accumulates the row / column sums of its own tile in the epilogue (per-tile partials, merged in a fixed order: the
//     result does not depend on the launch's timing or on the other proposals of the batch).
//   * one pass finds both label vectors (row arg-max complete per wave, column arg-max per 64-row slab + a small merge),
//   * one pass forms weight and pred.
// E is stored with a row stride of 2052 floats and logical column m at physical column m + 3, so that column 1 -- where the 128-wide
// GEMM tiles start -- is 16-byte aligned (the bg row / bg column are written by a small dot-product kernel).
// The operands are L2-normalised, multiplied by 2^10 and cut into fp16 hi / lo halves once (fm_prep_kernel), so the GEMM's main loop
// is copies + MFMA only (the generic kernel splits both operands on every k-step of every tile: 144 VALU instructions per thread).
#include "common.h"
#include "../../include/sam6d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float fm_f4 __attribute__((ext_vector_type(4)));

#define FM_C 256          // channels
#define FM_COL0 3         // physical column of logical column 0
#define FM_CHUNK 2048     // columns one label / assignment workgroup covers (32 per lane); np = 2048 (config 2) or 4096 (config 5)
#define FM_OPSCALE 1024.0f
// np = points per cloud (n - 1), a multiple of 2048.  Row stride of E: np + 4 floats; partial sums per row / column: 2 per 128-wide tile
// + the bg column / row; 64-row slabs in the label pass.
#define FM_LDE(np) ((np) + 4)
#define FM_SLOTS(np) (2 * ((np) / 128) + 1)
#define FM_SLABS(np) ((np) / 64)

// ---------------------------------------------------------------------------------------------------------------------
// F.normalize(dim=-1) (x / max(|x|, 1e-12), model_utils.py:141-142) * 2^10 -> fp16 hi / lo.  One wave per 256-channel row.
__global__ __launch_bounds__(256) void fm_prep_kernel(const float* __restrict__ f, long rows, _Float16* __restrict__ fh,
                                                      _Float16* __restrict__ fl) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const float4 v = *reinterpret_cast<const float4*>(f + row * FM_C + lane * 4);
  const float nrm = sqrtf(wave_sum_dpp((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)));
  const float d = fmaxf(nrm, 1e-12f);
  const float x[4] = {(v.x / d) * FM_OPSCALE, (v.y / d) * FM_OPSCALE, (v.z / d) * FM_OPSCALE, (v.w / d) * FM_OPSCALE};
  half4 hi, lo;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    _Float16 h, l;
    sam6d_split_f16(x[u], h, l);
    hi[u] = h;
    lo[u] = l;
  }
  *reinterpret_cast<half4*>(fh + row * FM_C + lane * 4) = hi;
  *reinterpret_cast<half4*>(fl + row * FM_C + lane * 4) = lo;
}

// ---------------------------------------------------------------------------------------------------------------------
// E tile (128 x 128) = exp2(acc * k1 - k2), acc = sum_k f1[n][k] f2[m][k] (fp16 x3 split products, fp32 accumulate), for the dense
// points n, m = 1 .. 2048 of proposal b.  4 waves in a 2 x 2 grid, each 64 x 64 (2 x 2 tiles of v_mfma_f32_32x32x16_f16).
// Operand staging: 16-byte global loads of the pre-split halves -> ds_write_b128, next K chunk prefetched into registers.
// (Measured and removed, round 3: staging by LDS-DMA -- 1 KiB pieces of 16 rows x 64 B with a per-lane swizzled source chunk so that the
// linear LDS image is conflict-free, two 32 KB slots, one barrier per chunk: the whole fine match 0.575 ms against 0.551 ms; two
// workgroups per CU instead of three to four, and eight DMA pieces of ~45 issue cycles per wave and chunk.)
// (Measured and removed, round 4: 256 x 128 tiles -- eight waves in a 4 x 2 grid, 70 KB of dynamic LDS, a quarter less operand traffic from
// the L2 per output.  At 128 registers (two workgroups per CU) the kernel spills 9 VGPRs, at 151 only one
// workgroup of eight waves fits a CU; bit-identical results, the fine-match pipeline 0.592 ms against 0.537 ms with the 128 x 128 tile.)
#define FM_BK 32
#define FM_LD 40  // halves per LDS row (80 B): the ds_read_b128 fragment reads of 16 consecutive rows are conflict-free
__global__ __launch_bounds__(256) void fm_sim_kernel(const _Float16* __restrict__ fh, const _Float16* __restrict__ fl, int B,
                                                     float k1, float k2, float* __restrict__ E, float* __restrict__ rowpart,
                                                     float* __restrict__ colpart, int half, int np) {
  __shared__ __attribute__((aligned(16))) _Float16 smem[4 * 128 * FM_LD];  // 40 KB; reused by the epilogue's transpose slabs
  _Float16* Ah = smem;
  _Float16* Al = Ah + 128 * FM_LD;
  _Float16* Bh = Al + 128 * FM_LD;
  _Float16* Bl = Bh + 128 * FM_LD;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // proposal z's tiles get the workgroup ids congruent to z mod 8: one XCD (one L2) reads that proposal's operands
  const int nt = np / 128, per = nt * nt;
  const int g = blockIdx.x / (8 * per), r8 = blockIdx.x % (8 * per);
  const int b = g * 8 + (r8 & 7);
  if (b >= B) return;
  const int x = r8 >> 3, tm = x % nt, tn = x / nt;
  const long n = np + 1;
  const int lde = FM_LDE(np), slots = FM_SLOTS(np);
  const _Float16* a_h = fh + ((long)b * n + 1 + 128 * tm) * FM_C;        // scene cloud b, rows 1 + 128 tm ..
  const _Float16* a_l = fl + ((long)b * n + 1 + 128 * tm) * FM_C;
  const _Float16* w_h = fh + ((long)(B + b) * n + 1 + 128 * tn) * FM_C;  // template cloud B + b
  const _Float16* w_l = fl + ((long)(B + b) * n + 1 + 128 * tn) * FM_C;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // staging: thread t copies rows (t >> 2) and (t >> 2) + 64, halves (t & 3) * 8 .. +8 of the chunk, for each of the 4 planes
  const int sr = t >> 2, sk = (t & 3) * 8;
  half8 v[8];
  auto fetch = [&](int k0) {
    v[0] = *reinterpret_cast<const half8*>(a_h + (long)sr * FM_C + k0 + sk);
    v[1] = *reinterpret_cast<const half8*>(a_h + (long)(sr + 64) * FM_C + k0 + sk);
    v[2] = *reinterpret_cast<const half8*>(a_l + (long)sr * FM_C + k0 + sk);
    v[3] = *reinterpret_cast<const half8*>(a_l + (long)(sr + 64) * FM_C + k0 + sk);
    v[4] = *reinterpret_cast<const half8*>(w_h + (long)sr * FM_C + k0 + sk);
    v[5] = *reinterpret_cast<const half8*>(w_h + (long)(sr + 64) * FM_C + k0 + sk);
    v[6] = *reinterpret_cast<const half8*>(w_l + (long)sr * FM_C + k0 + sk);
    v[7] = *reinterpret_cast<const half8*>(w_l + (long)(sr + 64) * FM_C + k0 + sk);
  };
  const int fr = lane & 31, fk = lane >> 5;
  fetch(0);
  for (int k0 = 0; k0 < FM_C; k0 += FM_BK) {
    __syncthreads();
    *reinterpret_cast<half8*>(&Ah[sr * FM_LD + sk]) = v[0];
    *reinterpret_cast<half8*>(&Ah[(sr + 64) * FM_LD + sk]) = v[1];
    *reinterpret_cast<half8*>(&Al[sr * FM_LD + sk]) = v[2];
    *reinterpret_cast<half8*>(&Al[(sr + 64) * FM_LD + sk]) = v[3];
    *reinterpret_cast<half8*>(&Bh[sr * FM_LD + sk]) = v[4];
    *reinterpret_cast<half8*>(&Bh[(sr + 64) * FM_LD + sk]) = v[5];
    *reinterpret_cast<half8*>(&Bl[sr * FM_LD + sk]) = v[6];
    *reinterpret_cast<half8*>(&Bl[(sr + 64) * FM_LD + sk]) = v[7];
    __syncthreads();
    if (k0 + FM_BK < FM_C) fetch(k0 + FM_BK);
#pragma unroll
    for (int ks = 0; ks < FM_BK; ks += 16) {
      half8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const half8*>(&Ah[(wm + 32 * i + fr) * FM_LD + ks + 8 * fk]);
        al[i] = *reinterpret_cast<const half8*>(&Al[(wm + 32 * i + fr) * FM_LD + ks + 8 * fk]);
        bh[i] = *reinterpret_cast<const half8*>(&Bh[(wn + 32 * i + fr) * FM_LD + ks + 8 * fk]);
        bl[i] = *reinterpret_cast<const half8*>(&Bl[(wn + 32 * i + fr) * FM_LD + ks + 8 * fk]);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if (!half) {  // (launch-uniform) matmul mode 2 keeps the hi . hi product only
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
  }
  // ---- E = exp(att - c), in the accumulator registers
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = __builtin_amdgcn_exp2f(fmaf(acc[i][j][r], k1, -k2));
  // ---- column partial sums of this wave's 64 rows: lane (fr, fk) holds rows (r&3) + 8(r>>2) + 4 fk of column fr
  const int nrow0 = 128 * tm, mcol0 = 128 * tn;  // point indices (n - 1, m - 1) of the tile origin
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    s += xor32_f32(s);
    if (fk == 0) colpart[((size_t)b * slots + 2 * tm + (wave >> 1)) * np + mcol0 + wn + 32 * j + fr] = s;
  }
  // ---- row partial sums over this wave's 64 columns
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float s = acc[i][0][r] + acc[i][1][r];
      s = row16_sum_dpp(s);
      s += xor16_f32(s);
      if (fr == 0) rowpart[((size_t)b * np + nrow0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * fk) * slots + 2 * tn + (wave & 1)] = s;
    }
  // ---- E store: each wave transposes its 32 x 64 slabs through LDS so that a lane owns 4 consecutive columns (16-byte stores)
  constexpr int WC = 64, SLD = WC + 4, LPR = WC / 4, RPP = 64 / LPR, NP = 32 / RPP;
  __syncthreads();
  float* slab = reinterpret_cast<float*>(smem) + wave * (32 * SLD);
  const int rr0 = lane / LPR, c4 = (lane % LPR) * 4;
  float* Eb = E + (size_t)b * (np + 1) * lde + FM_COL0;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[((r & 3) + 8 * (r >> 2) + 4 * fk) * SLD + j * 32 + fr] = acc[i][j][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int it = 0; it < NP; ++it) {
      const int rr = it * RPP + rr0;
      const int nn = 1 + nrow0 + wm + 32 * i + rr, mm = 1 + mcol0 + wn + c4;
      // non-temporal: E (537 MB per step) is read back only by the label / assignment passes, long after it has left the L2 -- kept out
      // of the cache it stops evicting the GEMM operands (4 MB per proposal = one XCD's L2) the neighbouring tiles are about to re-read
      __builtin_nontemporal_store(*reinterpret_cast<const fm_f4*>(&slab[rr * SLD + c4]), reinterpret_cast<fm_f4*>(Eb + (size_t)nn * lde + mm));
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// bg row and bg column: E[0][m] (m = 0 .. np) and E[n][0] (n = 1 .. np), one wave per dot product over the same scaled
// hi + lo operands the GEMM multiplies; 8 outputs per wave (64 per wave left each wave a chain of 64 dependent row loads: 66 us).
#define FM_BG_PER_WAVE 8
__global__ __launch_bounds__(256) void fm_bg_kernel(const _Float16* __restrict__ fh, const _Float16* __restrict__ fl, int B, float k1,
                                                    float k2, float* __restrict__ E, int np) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long n = np + 1;
  const int lde = FM_LDE(np);
  float* Eb = E + (size_t)b * n * lde + FM_COL0;
  auto rowv = [&](long row, float* x) {
    const half4 h = *reinterpret_cast<const half4*>(fh + row * FM_C + lane * 4);
    const half4 l = *reinterpret_cast<const half4*>(fl + row * FM_C + lane * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) x[u] = (float)h[u] + (float)l[u];
  };
  float s0[4], t0[4];
  rowv((long)b * n, s0);        // bg token of the scene cloud
  rowv((long)(B + b) * n, t0);  // bg token of the template cloud
  for (int o = (blockIdx.x * 4 + wave) * FM_BG_PER_WAVE, e = o + FM_BG_PER_WAVE; o < e; ++o) {
    if (o >= 2 * np + 1) break;
    float y[4];
    float d;
    if (o <= np) {  // E[0][m], m = o
      rowv((long)(B + b) * n + o, y);
      d = fmaf(s0[3], y[3], fmaf(s0[2], y[2], fmaf(s0[1], y[1], s0[0] * y[0])));
    } else {           // E[n][0], n = o - 2048
      rowv((long)b * n + (o - np), y);
      d = fmaf(t0[3], y[3], fmaf(t0[2], y[2], fmaf(t0[1], y[1], t0[0] * y[0])));
    }
    d = wave_sum_dpp(d);
    if (lane == 0) {
      const float ev = __builtin_amdgcn_exp2f(fmaf(d, k1, -k2));
      if (o <= np) Eb[o] = ev; else Eb[(size_t)(o - np) * lde] = ev;
    }
  }
}

// rsum[b][n] = sum_m E[n][m], csum[b][m] = sum_n E[n][m] from the tile partials + the bg column / row, in a fixed order.
// grid (9, B): threads 0 .. 2047 of the x range own a row AND a column; entry 2048 -> row 0 / column 0 (read from E).
__global__ __launch_bounds__(256) void fm_merge_sums_kernel(const float* __restrict__ rowpart, const float* __restrict__ colpart,
                                                            const float* __restrict__ E, float* __restrict__ rsum,
                                                            float* __restrict__ csum, int np) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  const int lde = FM_LDE(np), slots = FM_SLOTS(np);
  const float* Eb = E + (size_t)b * (np + 1) * lde + FM_COL0;
  if (i < np) {
    const float* rp = rowpart + ((size_t)b * np + i) * slots;
    float s = 0.f;
#pragma unroll 8
    for (int k = 0; k < slots - 1; ++k) s += rp[k];  // (slots - 1 = 2 np / 128: a multiple of 8; same ascending order)
    rsum[(size_t)b * (np + 1) + 1 + i] = s + Eb[(size_t)(1 + i) * lde];  // + E[n][0]
    float c = 0.f;
#pragma unroll 8
    for (int k = 0; k < slots - 1; ++k) c += colpart[((size_t)b * slots + k) * np + i];
    csum[(size_t)b * (np + 1) + 1 + i] = c + Eb[1 + i];                    // + E[0][m]
  }
  if (blockIdx.x == np / 256) {  // row 0 / column 0 (bg token): np + 1 terms each, summed by the whole workgroup in a fixed order
    __shared__ float red[2][256];
    const int t = threadIdx.x;
    float s = 0.f, c = 0.f;
    for (int m = t; m <= np; m += 256) s += Eb[m];
    for (int nn = t; nn <= np; nn += 256) c += Eb[(size_t)nn * lde];
    red[0][t] = s;
    red[1][t] = c;
    __syncthreads();
    if (t < 2) {
      float a = 0.f;
      for (int k = 0; k < 256; ++k) a += red[t][k];
      (t == 0 ? rsum : csum)[(size_t)b * (np + 1)] = a;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Labels.  S[n][m] = (E irs[n]) (E ics[m]).  grid (32 slabs, B); slab s = rows 1 + 64 s .. 64 s + 64 (slab 0 also row 0, for the
// column arg-max only); wave w takes rows w, w + 4, ... of the slab in ascending order.  A lane owns columns 1 + 4 (lane + 64 g) + j
// (g < 8, j < 4): 16-byte loads.  First maximum wins everywhere (torch.max; strict > in ascending index order, lower index on ties).
__device__ __forceinline__ void fm_better(float& bv, int& bi, float v, int i) {
  if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

__global__ __launch_bounds__(256) void fm_labels_kernel(const float* __restrict__ E, const float* __restrict__ rsum,
                                                        const float* __restrict__ csum, int* __restrict__ label1,
                                                        float* __restrict__ pbest, int* __restrict__ pidx, int np,
                                                        float* __restrict__ rowbest, int* __restrict__ rowidx) {
  // blockIdx.z = column chunk (2048 columns): with one chunk the row labels are final here, with more they are candidates per chunk
  // (rowbest / rowidx [chunk][b][np]) merged by fm_merge_rows_kernel
  __shared__ float sv[3][FM_CHUNK];
  __shared__ int si[3][FM_CHUNK];
  const int b = blockIdx.y, slab = blockIdx.x, chunk = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lde = FM_LDE(np), slabs = FM_SLABS(np), c0 = chunk * FM_CHUNK;
  const float* Eb = E + (size_t)b * (np + 1) * lde + FM_COL0 + c0;
  const float* rs = rsum + (size_t)b * (np + 1);
  const float* cs = csum + (size_t)b * (np + 1) + c0;
  float ics[32], cb[32];
  int ci[32];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 1 + 4 * (lane + 64 * g));
    ics[4 * g + 0] = __frcp_rn(c4.x); ics[4 * g + 1] = __frcp_rn(c4.y); ics[4 * g + 2] = __frcp_rn(c4.z); ics[4 * g + 3] = __frcp_rn(c4.w);
  }
#pragma unroll
  for (int k = 0; k < 32; ++k) { cb[k] = -INFINITY; ci[k] = 0x7fffffff; }
  const float ics0 = __frcp_rn(csum[(size_t)b * (np + 1)]);
  auto do_row = [&](int nn, bool want_row_label) {
    const float irs = __frcp_rn(rs[nn]);
    const float* er = Eb + (size_t)nn * lde;
    float4 e[8];
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const fm_f4 x = __builtin_nontemporal_load(reinterpret_cast<const fm_f4*>(er + 1 + 4 * (lane + 64 * g)));  // streamed once
      e[g] = make_float4(x[0], x[1], x[2], x[3]);
    }
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    if (lane == 0 && chunk == 0) {  // the bg column: a candidate of the row arg-max only
      const float e0 = er[0];
      bv = (e0 * irs) * (e0 * ics0);
      bi = 0;
    }
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const float ev[4] = {e[g].x, e[g].y, e[g].z, e[g].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float v = (ev[j] * irs) * (ev[j] * ics[4 * g + j]);
        if (v > bv) { bv = v; bi = c0 + 1 + 4 * (lane + 64 * g) + j; }
        if (v > cb[4 * g + j]) { cb[4 * g + j] = v; ci[4 * g + j] = nn; }
      }
    }
    if (want_row_label) {
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        fm_better(bv, bi, ov, oi);
      }
      if (lane == 0) {
        if (gridDim.z == 1) label1[(size_t)b * np + nn - 1] = bi;
        else {
          rowbest[((size_t)chunk * gridDim.y + b) * np + nn - 1] = bv;
          rowidx[((size_t)chunk * gridDim.y + b) * np + nn - 1] = bi;
        }
      }
    }
  };
  // (a lane visits its columns in ascending order, g-major, so `v > bv` keeps its first maximum; across lanes fm_better prefers the
  // lower column on ties)
  if (slab == 0 && wave == 0) do_row(0, false);
  for (int k = 0; k < 16; ++k) do_row(1 + 64 * slab + wave + 4 * k, true);
  // combine the four waves' column candidates (lowest row wins ties), then one partial per slab
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int m = 4 * (lane + 64 * (k >> 2)) + (k & 3);
      sv[wave - 1][m] = cb[k];
      si[wave - 1][m] = ci[k];
    }
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int k = 0; k < 32; ++k) {
      const int m = 4 * (lane + 64 * (k >> 2)) + (k & 3);
      float bv = cb[k];
      int bi = ci[k];
#pragma unroll
      for (int w = 0; w < 3; ++w) fm_better(bv, bi, sv[w][m], si[w][m]);
      pbest[((size_t)b * slabs + slab) * np + c0 + m] = bv;
      pidx[((size_t)b * slabs + slab) * np + c0 + m] = bi;
    }
  }
}

__global__ __launch_bounds__(256) void fm_merge_labels_kernel(const float* __restrict__ pbest, const int* __restrict__ pidx,
                                                              int* __restrict__ label2, int np) {
  const int b = blockIdx.y, m = blockIdx.x * 256 + threadIdx.x;
  if (m >= np) return;
  const int slabs = FM_SLABS(np);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int s = 0; s < slabs; ++s) {  // slabs in ascending row order: strict > keeps the first maximum
    const float v = pbest[((size_t)b * slabs + s) * np + m];
    if (v > bv) { bv = v; bi = pidx[((size_t)b * slabs + s) * np + m]; }
  }
  label2[(size_t)b * np + m] = (bi == 0x7fffffff) ? 0 : bi;
}

// np > 2048: row arg-max over the column chunks (ascending column order: strict > keeps the first maximum)
__global__ __launch_bounds__(256) void fm_merge_rows_kernel(const float* __restrict__ rowbest, const int* __restrict__ rowidx, int B,
                                                            int np, int chunks, int* __restrict__ label1) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int c = 0; c < chunks; ++c) {
    const float v = rowbest[((size_t)c * B + b) * np + i];
    if (v > bv) { bv = v; bi = rowidx[((size_t)c * B + b) * np + i]; }
  }
  label1[(size_t)b * np + i] = (bi == 0x7fffffff) ? 0 : bi;
}

// ---------------------------------------------------------------------------------------------------------------------
// Assignment: weight[n] = sum_m A[n][m], pred[n] = sum_m A[n][m] pts2[m] / (weight + 1e-6), A = S [l1[n] > 0] [l2[m] > 0].
// Same decomposition as the label pass; rows whose label is the bg token contribute nothing and are not read.
__global__ __launch_bounds__(256) void fm_assign_kernel(const float* __restrict__ E, const float* __restrict__ rsum,
                                                        const float* __restrict__ csum, const int* __restrict__ label1,
                                                        const int* __restrict__ label2, const float* __restrict__ pts2,
                                                        float* __restrict__ pred, float* __restrict__ weight, int np,
                                                        float* __restrict__ part) {
  // blockIdx.z = column chunk; with more than one chunk the four sums of a row go to part[chunk][b][np][4] (fm_assign_merge_kernel)
  const int b = blockIdx.y, slab = blockIdx.x, chunk = blockIdx.z, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lde = FM_LDE(np), c0 = chunk * FM_CHUNK;
  const float* Eb = E + (size_t)b * (np + 1) * lde + FM_COL0 + c0;
  const float* rs = rsum + (size_t)b * (np + 1);
  const float* cs = csum + (size_t)b * (np + 1) + c0;
  const int* l2 = label2 + (size_t)b * np + c0;
  const float* p2 = pts2 + ((size_t)b * np + c0) * 3;
  float gc[32], px[32], py[32], pz[32];
#pragma unroll
  for (int g = 0; g < 8; ++g) {
    const int m0 = 4 * (lane + 64 * g);
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 1 + m0);
    const int4 l4 = *reinterpret_cast<const int4*>(l2 + m0);
    gc[4 * g + 0] = l4.x > 0 ? __frcp_rn(c4.x) : 0.f;
    gc[4 * g + 1] = l4.y > 0 ? __frcp_rn(c4.y) : 0.f;
    gc[4 * g + 2] = l4.z > 0 ? __frcp_rn(c4.z) : 0.f;
    gc[4 * g + 3] = l4.w > 0 ? __frcp_rn(c4.w) : 0.f;
    const float4 q0 = *reinterpret_cast<const float4*>(p2 + 3 * m0);
    const float4 q1 = *reinterpret_cast<const float4*>(p2 + 3 * m0 + 4);
    const float4 q2 = *reinterpret_cast<const float4*>(p2 + 3 * m0 + 8);
    px[4 * g + 0] = q0.x; py[4 * g + 0] = q0.y; pz[4 * g + 0] = q0.z;
    px[4 * g + 1] = q0.w; py[4 * g + 1] = q1.x; pz[4 * g + 1] = q1.y;
    px[4 * g + 2] = q1.z; py[4 * g + 2] = q1.w; pz[4 * g + 2] = q2.x;
    px[4 * g + 3] = q2.y; py[4 * g + 3] = q2.z; pz[4 * g + 3] = q2.w;
  }
  for (int k = 0; k < 16; ++k) {
    const int nn = 1 + 64 * slab + wave + 4 * k;
    const size_t w = (size_t)b * np + nn - 1;
    float sa = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
    if (label1[w] > 0) {  // (wave-uniform)
      const float irs = __frcp_rn(rs[nn]);
      const float* er = Eb + (size_t)nn * lde;
      float4 e[8];
#pragma unroll
      for (int g = 0; g < 8; ++g) {
      const fm_f4 x = __builtin_nontemporal_load(reinterpret_cast<const fm_f4*>(er + 1 + 4 * (lane + 64 * g)));  // streamed once
      e[g] = make_float4(x[0], x[1], x[2], x[3]);
    }
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        const float ev[4] = {e[g].x, e[g].y, e[g].z, e[g].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v = (ev[j] * irs) * (ev[j] * gc[4 * g + j]);
          sa += v;
          sx = fmaf(v, px[4 * g + j], sx);
          sy = fmaf(v, py[4 * g + j], sy);
          sz = fmaf(v, pz[4 * g + j], sz);
        }
      }
      sa = wave_sum(sa); sx = wave_sum(sx); sy = wave_sum(sy); sz = wave_sum(sz);
    }
    if (lane == 0) {
      if (gridDim.z == 1) {
        const float den = sa + 1e-6f;
        weight[w] = sa;
        pred[w * 3] = sx / den;
        pred[w * 3 + 1] = sy / den;
        pred[w * 3 + 2] = sz / den;
      } else {
        *reinterpret_cast<float4*>(part + (((size_t)chunk * gridDim.y + b) * np + nn - 1) * 4) = make_float4(sa, sx, sy, sz);
      }
    }
  }
}

__global__ __launch_bounds__(256) void fm_assign_merge_kernel(const float* __restrict__ part, int B, int np, int chunks,
                                                              float* __restrict__ pred, float* __restrict__ weight) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= np) return;
  float sa = 0.f, sx = 0.f, sy = 0.f, sz = 0.f;
  for (int c = 0; c < chunks; ++c) {  // chunks in ascending column order
    const float4 p = *reinterpret_cast<const float4*>(part + (((size_t)c * B + b) * np + i) * 4);
    sa += p.x; sx += p.y; sy += p.z; sz += p.w;
  }
  const size_t w = (size_t)b * np + i;
  const float den = sa + 1e-6f;
  weight[w] = sa;
  pred[w * 3] = sx / den;
  pred[w * 3 + 1] = sy / den;
  pred[w * 3 + 2] = sz / den;
}

static size_t fm_align(size_t x) { return (x + 255) & ~(size_t)255; }

static size_t fm_workspace_bytes(int B, int np) {
  const size_t rows = (size_t)2 * B * (np + 1);
  const int chunks = np / FM_CHUNK;
  size_t s = 0;
  s += 2 * fm_align(rows * FM_C * 2);                               // fh, fl
  s += fm_align((size_t)B * (np + 1) * FM_LDE(np) * 4 + 64);        // E
  s += 2 * fm_align((size_t)B * np * FM_SLOTS(np) * 4);             // rowpart, colpart
  s += 2 * fm_align((size_t)B * (np + 1) * 4);                      // rsum, csum
  s += 2 * fm_align((size_t)B * FM_SLABS(np) * np * 4);             // pbest, pidx
  if (chunks > 1) s += 2 * fm_align((size_t)chunks * B * np * 4) + fm_align((size_t)chunks * B * np * 16);  // row candidates, sums
  return s;
}

extern "C" size_t sam6d_fine_match_workspace_bytes(int B) { return fm_workspace_bytes(B, 2048); }
extern "C" size_t sam6d_fine_match_workspace_bytes_n(int B, int n) {
  return (n > 1 && (n - 1) % FM_CHUNK == 0) ? fm_workspace_bytes(B, n - 1) : 0;
}

static int fm_run(const float* f, const _Float16* fh_in, const _Float16* fl_in, int B, int n, float temp, const float* pts2,
                  int* label1, int* label2, float* pred, float* weight, void* ws, size_t ws_bytes, void* stream);
extern "C" int sam6d_fine_match(const float* f, int B, int n, float temp, const float* pts2, int* label1, int* label2, float* pred,
                                float* weight, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(f, "fine_match: null pointer");
  return fm_run(f, nullptr, nullptr, B, n, temp, pts2, label1, label2, pred, weight, ws, ws_bytes, stream);
}
extern "C" int sam6d_fine_match_split(const void* fh, const void* fl, int B, int n, float temp, const float* pts2, int* label1,
                                      int* label2, float* pred, float* weight, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(fh && fl && ((((size_t)fh) | ((size_t)fl)) & 15) == 0, "fine_match_split: fh / fl must be 16-byte aligned pointers");
  return fm_run(nullptr, (const _Float16*)fh, (const _Float16*)fl, B, n, temp, pts2, label1, label2, pred, weight, ws, ws_bytes, stream);
}

// f: the out_proj features (fm_prep_kernel normalises and splits them into the workspace), or fh_in / fl_in: the split halves as
// sam6d_linear_norm_split leaves them
static int fm_run(const float* f, const _Float16* fh_in, const _Float16* fl_in, int B, int n, float temp, const float* pts2,
                  int* label1, int* label2, float* pred, float* weight, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(pts2 && label1 && label2 && pred && weight && ws, "fine_match: null pointer");
  SAM6D_REQUIRE(n == 2049 || n == 4097, "fine_match: built for n = 2049 or 4097 tokens per cloud (got %d)", n);
  SAM6D_REQUIRE(B >= 0 && B <= 4096 && temp > 0.f, "fine_match: bad sizes");
  const int np = n - 1, chunks = np / FM_CHUNK, lde = FM_LDE(np), slots = FM_SLOTS(np), slabs = FM_SLABS(np), nt = np / 128;
  SAM6D_REQUIRE(ws_bytes >= fm_workspace_bytes(B, np), "fine_match: workspace too small");
  SAM6D_REQUIRE((((size_t)ws | (size_t)f | (size_t)pts2 | (size_t)label2) & 15) == 0, "fine_match: pointers must be 16-byte aligned");
  SAM6D_REQUIRE(f || (fh_in && fl_in), "fine_match: no features");
  SAM6D_REQUIRE((long)cdiv(B, 8) * 8 * nt * nt < 2147483647L, "fine_match: too many tiles for one launch");
  if (B == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t rows = (size_t)2 * B * n;
  unsigned char* p = (unsigned char*)ws;
  const _Float16* fh = fh_in ? fh_in : (_Float16*)p; p += fm_align(rows * FM_C * 2);
  const _Float16* fl = fl_in ? fl_in : (_Float16*)p; p += fm_align(rows * FM_C * 2);
  float* E = (float*)p; p += fm_align((size_t)B * n * lde * 4 + 64);
  float* rowpart = (float*)p; p += fm_align((size_t)B * np * slots * 4);
  float* colpart = (float*)p; p += fm_align((size_t)B * np * slots * 4);
  float* rsum = (float*)p; p += fm_align((size_t)B * n * 4);
  float* csum = (float*)p; p += fm_align((size_t)B * n * 4);
  float* pbest = (float*)p; p += fm_align((size_t)B * slabs * np * 4);
  int* pidx = (int*)p; p += fm_align((size_t)B * slabs * np * 4);
  float* rowbest = nullptr; int* rowidx = nullptr; float* part = nullptr;
  if (chunks > 1) {
    rowbest = (float*)p; p += fm_align((size_t)chunks * B * np * 4);
    rowidx = (int*)p; p += fm_align((size_t)chunks * B * np * 4);
    part = (float*)p;
  }
  // E = exp(att - c) = exp2(acc * k1 - k2): acc carries the operand scale 2^20, att = acc / (2^20 temp), c = 1 / temp
  const float log2e = 1.4426950408889634f;
  const float k1 = log2e / (FM_OPSCALE * FM_OPSCALE * temp), k2 = log2e / temp;
  if (!fh_in) hipLaunchKernelGGL(fm_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, f, (long)rows, (_Float16*)fh, (_Float16*)fl);
  hipLaunchKernelGGL(fm_sim_kernel, dim3((unsigned)(cdiv(B, 8) * 8 * nt * nt)), dim3(256), 0, s, fh, fl, B, k1, k2, E, rowpart, colpart,
                     sam6d_half_for(3), np);
  hipLaunchKernelGGL(fm_bg_kernel, dim3(cdiv(2 * np + 1, 4 * FM_BG_PER_WAVE), B), dim3(256), 0, s, fh, fl, B, k1, k2, E, np);
  hipLaunchKernelGGL(fm_merge_sums_kernel, dim3(np / 256 + 1, B), dim3(256), 0, s, rowpart, colpart, E, rsum, csum, np);
  hipLaunchKernelGGL(fm_labels_kernel, dim3(slabs, B, chunks), dim3(256), 0, s, E, rsum, csum, label1, pbest, pidx, np, rowbest, rowidx);
  hipLaunchKernelGGL(fm_merge_labels_kernel, dim3(np / 256, B), dim3(256), 0, s, pbest, pidx, label2, np);
  if (chunks > 1) hipLaunchKernelGGL(fm_merge_rows_kernel, dim3(np / 256, B), dim3(256), 0, s, rowbest, rowidx, B, np, chunks, label1);
  hipLaunchKernelGGL(fm_assign_kernel, dim3(slabs, B, chunks), dim3(256), 0, s, E, rsum, csum, label1, label2, pts2, pred, weight, np,
                     part);
  if (chunks > 1) hipLaunchKernelGGL(fm_assign_merge_kernel, dim3(np / 256, B), dim3(256), 0, s, part, B, np, chunks, pred, weight);
  SAM6D_LAUNCH_CHECK("fine_match");
}
