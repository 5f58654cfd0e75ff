// Fused transformer-block kernels for 256-channel token rows: a tile of 128 tokens stays on chip from the block's input to
// its second LayerNorm.
//
//   mode 0 -- the tail every attention layer of PEM/model/transformer.py ends with (AttentionLayer.forward :152-160 /
//             RPEAttentionLayer.forward :436-444 after the attention itself, then AttentionOutput :184-199):
//                 y   = LayerNorm(hidden . Wlin^T + b + x)
//                 out = LayerNorm(relu(y . Wexp^T + b) . Wsq^T + b + y)
//   mode 1 -- the whole dense LinearTransformerLayer of the sparse-to-dense lift (:532-622) on the 2048 dense tokens of a cloud:
//                 q = D . Wq^T + b;  phi(q) (focused ReLU kernel);  z_h = 1 / (phi(q)_h . ksum_h + 1e-6);
//                 hidden_h = (phi(q)_h . kv_h) z_h;  then mode 0 with x = D.
//             (k / v side: the 196 sparse tokens, reduced to kv^T and key sums by linattn.hip)
//
// Before: 8 GEMM launches + 2 LayerNorm + focus passes per dense layer = twelve passes over 134 MB tensors (~3.2 GB of HBM traffic
// per layer); now D is read once and D' written once (268 MB).
//
// Layout: one workgroup = 8 waves (two per SIMD), one wave = 16 tokens, and every product is computed TRANSPOSED:
//     Y^T (out-channel x token) = W (out-channel x k) . X^T (k x token)          v_mfma_f32_16x16x32_f16
// so the weights are the A operand (streamed through LDS by LDS-DMA, shared by the eight waves) and a wave's activations are the B
// operand, held in registers for the whole K extent (8 k-steps x {hi, lo} x 4 VGPRs = 64 VGPRs).  In the 16x16 accumulator a lane
// (token = lane & 15, g = lane >> 4) holds out-channels 4g .. 4g+3 of ITS OWN token; two such tiles give the 8 values the B operand of
// the next product wants from that lane, up to a fixed permutation inside every 32-channel block (slot 8g + e <-> channel
// 16 (e >> 2) + 4g + (e & 3)), which is applied to the weights once, when they are packed.  Between two products there is only the
// row-wise epilogue (bias, focus / LayerNorm statistics: a lane's 64 values + the three lanes 16 / 32 / 48 away) and the fp16 hi/lo
// split -- no LDS slab, no transposition.  <= 256 VGPRs, so the epilogue of one wave runs beside the MFMAs of its SIMD partner.
//
// Arithmetic: fp16 x3 split precision like gemm_nt_h3_kernel (a_lo.b_hi + a_hi.b_lo + a_hi.b_hi, fp32 accumulate), made range-safe:
// every weight matrix carries a power-of-two scale chosen at pack time (max |w| -> [2^13, 2^14)), every activation row a power-of-two
// scale chosen from its own max (the FFN hidden row: from a bound computed at pack time, because its chunks accumulate into one
// accumulator); the scales are undone exactly in the epilogues.  No operand can overflow fp16, no lo half falls into subnormals.
#include "common.h"
#include "../../include/sam6d_hip.h"
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned tb_u4 __attribute__((ext_vector_type(4)));
typedef unsigned tb_u2 __attribute__((ext_vector_type(2)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// tokens per workgroup = 16 x waves: 64 (4 waves, 2-slot panel ring, two independent workgroups per CU) or 128 (8 waves, 4-slot ring)
// A panel = 32 weight rows x K (KS k-steps of 32): per row KS*32 hi halves | KS*32 lo halves, NO padding -- instead the 16-byte chunk
// c of row r is stored at chunk c ^ (r & 15), which makes the ds_read_b128 fragment reads (16 rows x 2 adjacent chunks per lane
// group) conflict-free.  The global image is stored swizzled, so a linear LDS-DMA copy reproduces it.
#define TB_ROWB(KS) ((KS) * 128)                // bytes of one weight row in a panel
#define TB_PBYTES(KS) (32 * TB_ROWB(KS))        // 32 768 (K=256), 16 384 (K=128), 8 192 (K=64): 32 / 16 / 8 DMA pieces of 1 KiB
#define TB_PANEL_BYTES TB_PBYTES(8)
#define TB_P256 TB_PBYTES(8)
#define TB_P128 TB_PBYTES(4)
#define TB_P64 TB_PBYTES(2)
#define TB_CHUNK_BYTES (4 * TB_P256 + 8 * TB_P128)  // one 128-wide FFN chunk: 4 expand panels + 8 squeeze panels
#define TB_Q_OFF (8 * TB_P256 + 4 * TB_CHUNK_BYTES) // proj_q panels follow the common part of the image
#define TB_IMAGE_BYTES(MODE) (TB_Q_OFF + ((MODE) ? 8 * TB_P256 : 0))
// constant vectors (floats)
#define TC_BQ 0
#define TC_ISP 256     // 1 / softplus(scale)
#define TC_BLIN 512
#define TC_G1 768
#define TC_BE1 1024
#define TC_BEXP 1280   // 512
#define TC_BSQ 1792
#define TC_G2 2048
#define TC_BE2 2304
#define TC_SC 2560     // inv_wq, inv_wlin, inv_wexp, inv_wsq_h (= 1 / (scale_wsq * scale_h)), scale_h
#define TC_N 2568

// slot p (0..31) of a 32-wide k-step <-> channel offset inside the step (see the header comment)
__host__ __device__ __forceinline__ int tb_slot_channel(int p) {
  const int g = p >> 3, e = p & 7;
  return 16 * (e >> 2) + 4 * g + (e & 3);
}

// ----------------------------------------------------------------------------------------------------- weight packing
// W (rows x ldw) fp32 -> panels of 32 rows x KS k-steps of 32 starting at column k0: per row [hi: KS*32 halves | lo: KS*32 halves |
// pad]; value = W * scale.
__global__ __launch_bounds__(256) void tb_pack_kernel(const float* __restrict__ W, long ldw, int k0, int KS, float scale,
                                                      unsigned char* __restrict__ dst) {
  const int panel = blockIdx.x;
  const int rowb = TB_ROWB(KS);
  unsigned char* out = dst + (size_t)panel * TB_PBYTES(KS);
  for (int i = threadIdx.x; i < 32 * KS * 32; i += 256) {
    const int m = i / (KS * 32), p = i % (KS * 32);
    const int col = k0 + 32 * (p >> 5) + tb_slot_channel(p & 31);
    const float v = W[(size_t)(panel * 32 + m) * ldw + col] * scale;
    _Float16 hi, lo;
    sam6d_split_f16(v, hi, lo);
    // logical half index -> 16-byte chunk (8 halves) -> swizzled chunk
    const int ch = p >> 3, cl = (KS * 32 + p) >> 3;
    _Float16* row = reinterpret_cast<_Float16*>(out + (size_t)m * rowb);
    row[((ch ^ (m & 15)) << 3) + (p & 7)] = hi;
    row[((cl ^ (m & 15)) << 3) + (p & 7)] = lo;
  }
}

extern "C" int sam6d_pack_panels(const float* W, long ldw, int rows, int k0, int ksteps, float scale, void* dst, void* stream) {
  SAM6D_REQUIRE(W && dst && rows > 0 && (rows % 32) == 0 && k0 >= 0 && (ksteps == 2 || ksteps == 4 || ksteps == 8) && scale > 0.f,
                "pack_panels: bad arguments (ksteps = K / 32 must be 2, 4 or 8)");
  hipLaunchKernelGGL(tb_pack_kernel, dim3(rows / 32), dim3(256), 0, (hipStream_t)stream, W, ldw, k0, ksteps, scale,
                     (unsigned char*)dst);
  SAM6D_LAUNCH_CHECK("pack_panels");
}

__device__ __forceinline__ float pow2_scale_for(float amax) {
  // power of two s with amax * s in [2^13, 2^14); 1 for zero / non-finite rows
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);  // amax = m 2^e, m in [0.5, 1)
  e = 14 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}

// per cloud: kv^T (4 heads x 64 d x 64 c) -> 8 panels of 32 d-rows x K = 64 (head h: panels 2h, 2h+1), every head scaled by its own
// power of two (from the head's max |kv|); inv[4 b + h] = 1 / scale.
__global__ __launch_bounds__(256) void tb_kv_pack_kernel(const float* __restrict__ kvT, unsigned char* __restrict__ dst,
                                                         float* __restrict__ inv) {
  __shared__ float red[4][4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* src = kvT + (size_t)b * 16384;
  float m[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int hd = 0; hd < 4; ++hd)
    for (int i = t; i < 4096; i += 256) m[hd] = fmaxf(m[hd], fabsf(src[hd * 4096 + i]));
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) {
    m[hd] = wave_max_dpp(m[hd]);
    if ((t & 63) == 0) red[hd][t >> 6] = m[hd];
  }
  __syncthreads();
  float scale[4];
#pragma unroll
  for (int hd = 0; hd < 4; ++hd) scale[hd] = pow2_scale_for(fmaxf(fmaxf(red[hd][0], red[hd][1]), fmaxf(red[hd][2], red[hd][3])));
  if (t < 4) inv[(size_t)b * 4 + t] = 1.0f / (t == 0 ? scale[0] : t == 1 ? scale[1] : t == 2 ? scale[2] : scale[3]);
  unsigned char* out = dst + (size_t)b * (8 * TB_P64);
  for (int i = t; i < 16384; i += 256) {
    const int hd = i >> 12, d = (i >> 6) & 63, p = i & 63;  // p: slot inside the 64-wide K
    const int c = 32 * (p >> 5) + tb_slot_channel(p & 31);
    const float sc = hd == 0 ? scale[0] : hd == 1 ? scale[1] : hd == 2 ? scale[2] : scale[3];
    const float v = src[(hd * 64 + d) * 64 + c] * sc;
    _Float16 hi, lo;
    sam6d_split_f16(v, hi, lo);
    const int mrow = d & 31;
    _Float16* row = reinterpret_cast<_Float16*>(out + (size_t)(2 * hd + (d >> 5)) * TB_P64 + (size_t)mrow * TB_ROWB(2));
    row[(((p >> 3) ^ (mrow & 15)) << 3) + (p & 7)] = hi;
    row[((((64 + p) >> 3) ^ (mrow & 15)) << 3) + (p & 7)] = lo;
  }
}

// ----------------------------------------------------------------------------------------------------- kv side in one launch
// phi(k) (the focused-ReLU kernel function, linattn.hip focus_row: same operations in the same order), kv^T = sum_j phi(k)_j^T v_j and
// the key sums of the 4 heads of one cloud, then the packed fp16 image of kv^T with the cloud's power-of-two scale -- what
// focus_k_kernel + kv_reduce_kernel + tb_kv_pack_kernel did in three launches (15 + 38 + 28 us at 64 clouds x 196 tokens).  One
// workgroup of 1024 threads per cloud: wave w stages key rows w, w + 16 of every 32-key tile (a wave = one 256-channel row: the focus
// norms are wave reductions), thread (head h = t >> 8, d = (t >> 2) & 63, c0 = 16 (t & 3)) owns kv^T[h][d][c0 .. c0 + 15] and
// accumulates over the keys in ascending order, exactly as kv_reduce_kernel does.
__device__ __forceinline__ float tbk_softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// Round 3: one workgroup per (cloud, HEAD) instead of per cloud (64 workgroups of 1024 threads on 64 of the 256 CUs, 68 us): 256
// threads own kv^T[h][d][c0 .. c0 + 15] of their head, the four waves stage the 16 key rows of a step (a wave = one 256-channel row:
// the focus norms need the whole row, only the head's 64 k and 64 v channels go to LDS), every head gets its own power-of-two scale.
// Same phi(k), same per-output accumulation order: kv^T and the key sums keep their bits.
// Round 4: THREADS per (cloud, head) workgroup is a template parameter.  With 256 threads (round 3) a CU held four waves, one per SIMD,
// and a step's chain -- four rows of phi(k) per wave with two wave reductions each, then 16 keys x (LDS reads + 16 fma) per thread --
// ran without anything to overlap with: 56 us, unchanged by deeper row prefetch (58.7 us with a four-step register ring: not a load-
// latency problem).  1024 threads: one row per wave and step, four outputs per thread, sixteen waves per CU.  The accumulation order
// of every output (keys ascending) does not depend on the thread that owns it: same bits.
template <int THREADS>
__global__ __launch_bounds__(THREADS) void tb_kv_fused_kernel(const float* __restrict__ kv, const float* __restrict__ scale, int J, long ld,
                                                              long sb, unsigned char* __restrict__ image, float* __restrict__ inv,
                                                              float* __restrict__ ksum) {
  constexpr int NW = THREADS / 64, RPW = 16 / NW;   // waves, key rows per wave and step
  constexpr int OPT = 4096 / THREADS, TPR = 64 / OPT;  // outputs per thread, threads per kv^T row
  __shared__ __attribute__((aligned(16))) float ks[16][64];
  __shared__ __attribute__((aligned(16))) float vs[16][68];
  __shared__ float red[NW];
  const int h = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* kb = kv + (size_t)b * sb;
  const float4 sc = *reinterpret_cast<const float4*>(scale + lane * 4);
  const float sp[4] = {tbk_softplus(sc.x), tbk_softplus(sc.y), tbk_softplus(sc.z), tbk_softplus(sc.w)};
  const int d = t / TPR, c0 = (t % TPR) * OPT;
  const bool mine = (lane >> 4) == h;  // this lane's 4 channels belong to head h
  float acc[OPT];
#pragma unroll
  for (int u = 0; u < OPT; ++u) acc[u] = 0.f;
  float ksacc = 0.f;  // threads t < 64 accumulate ksum[h][c = t]
  // this wave's key rows of a step (rows wave, wave + NW, ...) in a register ring of NS steps: the rows of step i + NS - 1 are requested
  // before step i is accumulated.  (NS = 4 measured: 58.7 against 56.4 us at 256 threads, 41.1 against 41.3 us at 1024 -- the loop is
  // bound by the instructions it issues, ~130 per key row for phi(k) with its eight IEEE divisions and ~130 per step and wave for the
  // accumulation, not by the loads.)
  constexpr int NS = 2;
  float4 xr[NS][RPW], vr[NS][RPW];
  auto fetch = [&](float4 (&xq)[RPW], float4 (&vq)[RPW], int j0) {
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
      const int j = j0 + NW * rr + wave;
      xq[rr] = vq[rr] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < J) {
        xq[rr] = *reinterpret_cast<const float4*>(kb + (size_t)j * ld + lane * 4);
        vq[rr] = *reinterpret_cast<const float4*>(kb + (size_t)j * ld + 256 + lane * 4);
      }
    }
  };
#pragma unroll
  for (int st = 0; st < NS - 1; ++st)
    if (st * 16 < J) fetch(xr[st], vr[st], st * 16);
  for (int jb = 0; jb < J; jb += NS * 16) {
#pragma unroll
    for (int st = 0; st < NS; ++st) {
      const int j0 = jb + st * 16;
      if (j0 < J) {  // (uniform)
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
          const int jr = NW * rr + wave, j = j0 + jr;
          float4 kx = make_float4(0.f, 0.f, 0.f, 0.f);
          const float4 vx = vr[st][rr];
          if (j < J) {
            const float4 x = xr[st][rr];
            float a[4] = {x.x, x.y, x.z, x.w}, c[4];
            float n1 = 0.f, n3 = 0.f;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              float v = (a[u] > 0.f ? a[u] : 0.f) + 1e-6f;
              v = v / sp[u];
              n1 += v * v;
              c[u] = (v * v) * v;
              n3 += c[u] * c[u];
            }
            n1 = sqrtf(wave_sum_dpp(n1));
            n3 = sqrtf(wave_sum_dpp(n3));
            kx = make_float4((c[0] / n3) * n1, (c[1] / n3) * n1, (c[2] / n3) * n1, (c[3] / n3) * n1);
          }
          if (mine) {
            *reinterpret_cast<float4*>(&ks[jr][(lane & 15) * 4]) = kx;
            *reinterpret_cast<float4*>(&vs[jr][(lane & 15) * 4]) = vx;
          }
        }
        __syncthreads();
        if (j0 + (NS - 1) * 16 < J) fetch(xr[(st + NS - 1) % NS], vr[(st + NS - 1) % NS], j0 + (NS - 1) * 16);
#pragma unroll 4
        for (int jj = 0; jj < 16; ++jj) {
          const float vv = vs[jj][d];
#pragma unroll
          for (int u = 0; u < OPT; ++u) acc[u] = fmaf(ks[jj][c0 + u], vv, acc[u]);
          if (t < 64) ksacc += ks[jj][t];
        }
      }
    }
  }
  if (t < 64) ksum[((size_t)b * 4 + h) * 64 + t] = ksacc;
  // the head's scale from max |kv^T_h|
  float m = 0.f;
#pragma unroll
  for (int u = 0; u < OPT; ++u) m = fmaxf(m, fabsf(acc[u]));
  m = wave_max_dpp(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  float mm = red[0];
#pragma unroll
  for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red[w]);
  const float s2 = pow2_scale_for(mm);
  if (t == 0) inv[(size_t)b * 4 + h] = 1.0f / s2;
  // image: head h, row d -> panel 2 h + (d >> 5), row m = d & 31; channel c sits in slot p = 32 (c >> 5) + 8 g + e of the 64-wide K
  // (tb_slot_channel inverted: g = (c >> 2) & 3, e = 4 ((c >> 4) & 1) + (c & 3))
  const int mrow = d & 31;
  _Float16* row = reinterpret_cast<_Float16*>(image + (size_t)b * (8 * TB_P64) + (size_t)(2 * h + (d >> 5)) * TB_P64 +
                                              (size_t)mrow * TB_ROWB(2));
#pragma unroll
  for (int u = 0; u < OPT; u += 2) {
    unsigned hi, lo;
    sam6d_split2_f16(acc[u] * s2, acc[u + 1] * s2, hi, lo);
    const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
    const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
#pragma unroll
    for (int w = 0; w < 2; ++w) {
      const int c = c0 + u + w, cc = c & 31;
      const int pslot = 32 * (c >> 5) + 8 * ((cc >> 2) & 3) + 4 * (cc >> 4) + (cc & 3);
      row[(((pslot >> 3) ^ (mrow & 15)) << 3) + (pslot & 7)] = h2[w];
      row[((((64 + pslot) >> 3) ^ (mrow & 15)) << 3) + (pslot & 7)] = l2[w];
    }
  }
}

extern "C" int sam6d_linattn_kv_image(const float* kv, const float* scale, int B, int J, long ld, long stride, void* image, float* inv,
                                      float* ksum, void* stream) {
  SAM6D_REQUIRE(kv && scale && image && inv && ksum && B >= 0 && B <= 65535 && J > 0 && ld >= 512 && (ld & 3) == 0 &&
                    (((size_t)kv | (size_t)scale) & 15) == 0,
                "linattn_kv_image: bad arguments (kv rows = 256 k | 256 v channels, 16-byte aligned)");
  if (B == 0) return 0;
  static int shape = -1;  // SAM6D_KV_THREADS = 256 / 512 / 1024 (A/B runs)
  if (shape < 0) {
    const char* e = getenv("SAM6D_KV_THREADS");
    shape = e ? atoi(e) : 1024;
  }
  if (shape == 256)
    hipLaunchKernelGGL(tb_kv_fused_kernel<256>, dim3(4, B), dim3(256), 0, (hipStream_t)stream, kv, scale, J, ld, stride,
                       (unsigned char*)image, inv, ksum);
  else if (shape == 512)
    hipLaunchKernelGGL(tb_kv_fused_kernel<512>, dim3(4, B), dim3(512), 0, (hipStream_t)stream, kv, scale, J, ld, stride,
                       (unsigned char*)image, inv, ksum);
  else
    hipLaunchKernelGGL(tb_kv_fused_kernel<1024>, dim3(4, B), dim3(1024), 0, (hipStream_t)stream, kv, scale, J, ld, stride,
                       (unsigned char*)image, inv, ksum);
  SAM6D_LAUNCH_CHECK("linattn_kv_image");
}

extern "C" int sam6d_linattn_kv_pack(const float* kvT, int B, void* image, float* inv, void* stream) {
  SAM6D_REQUIRE(kvT && image && inv && B >= 0, "linattn_kv_pack: bad arguments");
  if (B == 0) return 0;
  hipLaunchKernelGGL(tb_kv_pack_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, kvT, (unsigned char*)image, inv);
  SAM6D_LAUNCH_CHECK("linattn_kv_pack");
}

extern "C" long sam6d_token_block_image_bytes(int mode) { return mode ? TB_IMAGE_BYTES(1) : TB_IMAGE_BYTES(0); }
extern "C" long sam6d_linattn_kv_image_bytes(void) { return 8 * TB_P64; }

// ------------------------------------------------------------------------------------------------------- the kernel
// one 32-row panel = two 16-row out tiles; KS k-steps of 32.  The weight fragments of step s+1 are requested before the six MFMAs
// of step s are issued (the eight waves run in lock-step between the panel barriers, so nothing else hides the LDS latency).  The
// reads and their counted waits are inline assembly: left to the compiler, the machine scheduler sinks every read to just before its
// use and waits with lgkmcnt(0) (measured: 2.7x the MFMA time per panel).
typedef unsigned tb_u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ tb_u32x4 tb_lds128(unsigned addr) {
  tb_u32x4 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// wait until at most N LDS operations are outstanding; the fragments are in/out operands so that their uses stay behind the wait
template <int N>
__device__ __forceinline__ void tb_wait(tb_u32x4& a, tb_u32x4& b, tb_u32x4& c, tb_u32x4& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
// chunk (16 B) index of (plane P, step S, lane group g) in a row: P * KS * 4 + 4 S + g; its low four bits are XORed with the row:
// the lane-dependent part ((4 q + g) ^ row) << 4 for q = 0..3 sits in four address registers, the rest is an immediate offset
template <int KS, int P, int S>
struct TbChunk {
  static constexpr int c = P * KS * 4 + 4 * S;     // + g (g < 4 never carries out of the low two bits)
  static constexpr int q = (c & 15) >> 2;
  static constexpr int off = (c & ~15) * 16;
};
template <int KS, int S>
__device__ __forceinline__ void tb_load_step(const unsigned (&a)[4], tb_u32x4* f) {
  constexpr int R1 = 16 * TB_ROWB(KS);
  typedef TbChunk<KS, 0, S> H;
  typedef TbChunk<KS, 1, S> L;
  f[0] = tb_lds128<H::off>(a[H::q]);
  f[1] = tb_lds128<L::off>(a[L::q]);
  f[2] = tb_lds128<R1 + H::off>(a[H::q]);
  f[3] = tb_lds128<R1 + L::off>(a[L::q]);
}
// Fragment ring of TB_FD + 1 steps: the reads of step S + TB_FD are issued before the six MFMAs of step S (one step of MFMAs, 96
// cycles, does not cover the LDS latency when all eight waves of the CU stream b128 reads).
// Deeper rings were measured on the 197-token layers (round 3): two steps ahead within the same register budget 48.1 us against 48.3 us,
// three steps with a 4-slot panel ring and one wave per SIMD 52.5 us -- s_memtime stamps (scratch/tb_stamps.py) put a K = 256 panel at
// ~1420 cycles against 768 of MFMA issue with the waits for DMA and barrier at ~100 + ~160: the four waves of a workgroup each re-read
// the whole panel from LDS (64 LDS cycles per 96-cycle k-step), which a longer look-ahead does not change.
#define TB_FD 1
template <int KS, int S, int FD_>
__device__ __forceinline__ void tb_mma_step(f32x4& acc0, f32x4& acc1, const unsigned (&a)[4], const half8* __restrict__ xh,
                                            const half8* __restrict__ xl, tb_u32x4 (&f)[FD_ + 1][4], bool half) {
  constexpr int cur = S % (FD_ + 1);
#ifndef TB_ABL_NOREAD  // (scratch/abl_build.sh: timing-only builds without the fragment reads / the panel DMA / the panel barrier)
  if constexpr (S + FD_ < KS) tb_load_step<KS, S + FD_>(a, f[(S + FD_) % (FD_ + 1)]);
#endif
  constexpr int newer = (KS - 1 - S) < FD_ ? (KS - 1 - S) : FD_;  // steps requested after this one
  tb_wait<4 * newer>(f[cur][0], f[cur][1], f[cur][2], f[cur][3]);
  const half8 h0 = __builtin_bit_cast(half8, f[cur][0]), l0 = __builtin_bit_cast(half8, f[cur][1]);
  const half8 h1 = __builtin_bit_cast(half8, f[cur][2]), l1 = __builtin_bit_cast(half8, f[cur][3]);
  if (!half) {  // (launch-uniform) matmul mode 2 keeps the hi . hi product only
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l0, xh[S], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(l1, xh[S], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xl[S], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xl[S], acc1, 0, 0, 0);
  }
  acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h0, xh[S], acc0, 0, 0, 0);
  acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h1, xh[S], acc1, 0, 0, 0);
  if constexpr (S + 1 < KS) tb_mma_step<KS, S + 1, FD_>(acc0, acc1, a, xh, xl, f, half);
}
// `panel`: LDS byte address of the panel; rows fr (first out tile) and fr + 16 (second)
template <int KS, int FD = TB_FD>
__device__ __forceinline__ void tb_mma(f32x4& acc0, f32x4& acc1, unsigned panel, const half8* __restrict__ xh,
                                       const half8* __restrict__ xl, int fr, int fg, bool half) {
  const unsigned rowbase = panel + fr * TB_ROWB(KS);
  const unsigned a[4] = {rowbase + (((0 + fg) ^ fr) << 4), rowbase + (((4 + fg) ^ fr) << 4), rowbase + (((8 + fg) ^ fr) << 4),
                         rowbase + (((12 + fg) ^ fr) << 4)};
  tb_u32x4 f[FD + 1][4];
  tb_load_step<KS, 0>(a, f[0]);
  if constexpr (FD >= 2 && KS >= 2) tb_load_step<KS, 1>(a, f[1]);
  if constexpr (FD >= 3 && KS >= 3) tb_load_step<KS, 2>(a, f[2]);
  tb_mma_step<KS, 0, FD>(acc0, acc1, a, xh, xl, f, half);
}

// a token's channels live in the four lanes 16 apart (g = lane >> 4): reductions over the token
__device__ __forceinline__ float tok_max(float m) {
  m = fmaxf(m, xor16_f32(m));
  return fmaxf(m, xor32_f32(m));
}
__device__ __forceinline__ float tok_sum(float s) {
  s += xor16_f32(s);
  return s + xor32_f32(s);
}

// makes a register value opaque to the optimiser (no instruction): without it the compiler keeps the fp32 images of y's hi / lo
// halves, computed while splitting, alive across the whole FFN for the second residual (128 VGPRs -> scratch spills)
__device__ __forceinline__ void tb_opaque(half8& x) {
  tb_u32x4 t = __builtin_bit_cast(tb_u32x4, x);
  asm volatile("" : "+v"(t));
  x = __builtin_bit_cast(half8, t);
}

// split NT accumulator tiles (true units) into k-steps x{h,l}[t >> 1] (slot 4 (t & 1) + r); returns the row scale used
template <int NT>
__device__ __forceinline__ float tb_split_rows(const f32x4* v, half8* xh, half8* xl) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) m = fmaxf(m, fabsf(v[t][r]));
  const float sc = pow2_scale_for(tok_max(m));
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      unsigned hi, lo;
      sam6d_split2_f16(v[t][r] * sc, v[t][r + 1] * sc, hi, lo);
      const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
      const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
      xh[t >> 1][4 * (t & 1) + r] = h2[0];
      xh[t >> 1][4 * (t & 1) + r + 1] = h2[1];
      xl[t >> 1][4 * (t & 1) + r] = l2[0];
      xl[t >> 1][4 * (t & 1) + r + 1] = l2[1];
    }
  return sc;
}

// LayerNorm over the token's 256 channels, in place on 16 accumulator tiles; gamma / beta from LDS
__device__ __forceinline__ void tb_layernorm(f32x4* v, const float* __restrict__ g, const float* __restrict__ be, int fg, float eps) {
  float s = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) s += (v[t][0] + v[t][1]) + (v[t][2] + v[t][3]);
  const float mean = tok_sum(s) * (1.0f / 256.0f);
  float q = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      v[t][r] -= mean;
      q += v[t][r] * v[t][r];
    }
  const float rstd = 1.0f / sqrtf(tok_sum(q) * (1.0f / 256.0f) + eps);
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const float4 gg = *reinterpret_cast<const float4*>(g + 16 * t + 4 * fg);
    const float4 bb = *reinterpret_cast<const float4*>(be + 16 * t + 4 * fg);
    v[t][0] = v[t][0] * rstd * gg.x + bb.x;
    v[t][1] = v[t][1] * rstd * gg.y + bb.y;
    v[t][2] = v[t][2] * rstd * gg.z + bb.z;
    v[t][3] = v[t][3] * rstd * gg.w + bb.w;
  }
}

struct TbArgs {
  const float* in;        // mode 0: hidden (M,256); mode 1: D (B, I, 256)
  const float* resid;     // mode 0: x (M,256);      mode 1: unused (the residual is D)
  float* out;             // mode 0: (M,256);        mode 1: D' (B, I, 256)
  const unsigned char* wimg;
  const float* consts;
  const unsigned char* kvimg;  // mode 1: (B, 8 panels)
  const float* kvinv;          // mode 1: (B, 4) one inverse image scale per head
  const float* ksum;           // mode 1: (B, 256)
  long M;                 // mode 0: rows
  int I, row0, tiles_per_b;  // mode 1: rows per cloud, first row handled, tiles per cloud
  float eps;
  int half;               // 1: fp16 single product (matmul mode 2)
};

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E-1
template <int B, int E, class F>
__device__ __forceinline__ void tb_static_for(F&& f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    tb_static_for<B + 1, E>(f);
  }
}

// The panel sequence of a launch (MODE 1: proj_q x8, kv x8, then the common part; common: linear x8, 4 x {expand x4, squeeze x8}) and
// the number of 1 KiB DMA pieces EACH WAVE copies for panel i (8 waves): 4 (K = 256), 2 (K = 128), 1 (K = 64), 0 past the end.
template <int MODE, int WAVES>
struct TbSched {
  static constexpr int NPAN = (MODE ? 16 : 0) + 8 + 4 * 12;
  static constexpr int per_wave(int i) {
    if (i < 0 || i >= NPAN) return 0;
    if (MODE) {
      if (i < 8) return (32 + WAVES - 1) / WAVES;
      if (i < 16) return (8 + WAVES - 1) / WAVES;
      i -= 16;
    }
    if (i < 8) return (32 + WAVES - 1) / WAVES;
    return ((((i - 8) % 12) < 4 ? 32 : 16) + WAVES - 1) / WAVES;
  }
  static constexpr int per_issuer(int i, int nw) {  // pieces of panel i each of nw issuing waves copies
    return (i < 0 || i >= NPAN) ? 0 : (pieces(i) + nw - 1) / nw;
  }
  static constexpr int pieces(int i) {  // 1 KiB pieces of panel i
    if (MODE && i >= 8 && i < 16) return 8;
    const int k = MODE ? i - 16 : i;
    return (k >= 8 && ((k - 8) % 12) >= 4) ? 16 : 32;
  }
  // pieces of the panels I+1 .. I+NBUF-2 (in flight while panel I is awaited)
  template <int NBUF>
  static constexpr int in_flight(int i) {
    int n = 0;
    for (int k = 1; k <= NBUF - 2; ++k) n += per_wave(i + k);
    return n;
  }
};

#ifdef TB_STAMP  // diagnostic build only (scratch/stamp): per-wave s_memtime stamps around every panel's wait / barrier
#define TB_NSTAMP 256
__device__ unsigned long long tb_stamps[512 * 8 * TB_NSTAMP];
extern "C" int sam6d_tb_debug_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(tb_stamps), sizeof(unsigned long long) * 512 * 8 * TB_NSTAMP);
}
// (stamps go to LDS and are flushed at the end: a global store before a panel's s_waitcnt vmcnt would itself be waited for)
#define TB_ST(i) do { if (lane == 0 && wave < WAVES && (i) < TB_NSTAMP) st_lds[wave][(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TB_ST(i)
#endif

// LOADERS > 0: that many extra waves do nothing but the panel DMA (wait for their pieces, join the barrier, issue the next step), so
// the ~45 issue cycles of each 1 KiB piece (8 per wave and K = 256 panel) leave the computing waves' instruction streams; PSTEP = 2:
// a ring slot holds two consecutive panels and the barrier comes every other panel.  Both only where one workgroup per CU runs anyway
// (the 197-token layers: 197 workgroups on 256 CUs).
template <int MODE, int WAVES, int NBUF, int FD = TB_FD, int LOADERS = 0, int PSTEP = 1>
__global__ __launch_bounds__((WAVES + LOADERS) * 64, 2) void token_block_kernel(TbArgs a) {  // (2 waves per SIMD: <= 256 VGPR + AGPR)
  constexpr int TB_TOK = 16 * WAVES, TB_NBUF = NBUF;
  constexpr int NTHREADS = (WAVES + LOADERS) * 64;
  constexpr int SLOT_BYTES = PSTEP * TB_PANEL_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* pan = lds;                                                   // TB_NBUF slots of PSTEP panels
  float* cst = reinterpret_cast<float*>(lds + TB_NBUF * SLOT_BYTES);          // TC_N floats
  float* ksm = cst + TC_N;                                                    // 256 floats (mode 1)
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  typedef TbSched<MODE, WAVES> SCH;
  const bool half = a.half != 0;
#ifdef TB_STAMP
  __shared__ unsigned long long st_lds[WAVES][TB_NSTAMP];
  unsigned long long* st_base = tb_stamps + ((size_t)(blockIdx.x < 512 ? blockIdx.x : 0) * 8 + wave) * TB_NSTAMP;
#endif
  TB_ST(0);
  const unsigned pan_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)pan;

  // ---- which rows
  long row;       // global row of this lane's token
  bool valid;
  int b = 0;
  if (MODE == 0) {
    const long r0 = (long)blockIdx.x * TB_TOK + wave * 16 + fr;
    valid = r0 < a.M;
    row = valid ? r0 : a.M - 1;
  } else {
    b = blockIdx.x / a.tiles_per_b;
    const int tk = (blockIdx.x % a.tiles_per_b) * TB_TOK + wave * 16 + fr + a.row0;
    valid = tk < a.I;
    row = (long)b * a.I + (valid ? tk : a.I - 1);
  }

  const unsigned char* kvp = MODE ? a.kvimg + (size_t)b * (8 * TB_P64) : nullptr;
  static_assert(SCH::NPAN % PSTEP == 0, "whole steps");
  constexpr int NW = LOADERS ? LOADERS : WAVES;        // waves that issue the DMA
  const bool issuer = LOADERS ? wave >= WAVES : true;
  const int lw = LOADERS ? wave - WAVES : wave;        // index among the issuing waves
  // LDS-DMA of panel I into its ring slot: SCH::per_issuer(I, NW) pieces of 1 KiB per issuing wave (the global image IS the LDS image)
  auto dma = [&](auto IC) {
    constexpr int I = decltype(IC)::value;
    constexpr int NP = SCH::per_issuer(I, NW);
    if constexpr (NP > 0) {
      constexpr int K = MODE ? I - 16 : I;   // index in the common part (negative: proj_q / kv panels)
      const unsigned char* src;
      if constexpr (MODE && I < 8) src = a.wimg + TB_Q_OFF + (size_t)I * TB_P256;
      else if constexpr (MODE && I < 16) src = kvp + (size_t)(I - 8) * TB_P64;
      else if constexpr (K < 8) src = a.wimg + (size_t)K * TB_P256;
      else {
        constexpr int c = (K - 8) / 12, u = (K - 8) % 12;
        constexpr size_t base = 8 * (size_t)TB_P256 + c * (size_t)TB_CHUNK_BYTES;
        src = a.wimg + (u < 4 ? base + u * (size_t)TB_P256 : base + 4 * (size_t)TB_P256 + (u - 4) * (size_t)TB_P128);
      }
      unsigned char* dst = pan + ((I / PSTEP) % TB_NBUF) * SLOT_BYTES + (I % PSTEP) * TB_PANEL_BYTES;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int pc = lw + NW * k;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                         (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
      }
    }
  };
  // Start of panel I: this wave's pieces of panel I have landed once at most the pieces of the panels issued after it (I+1 ..
  // I+NBUF-2; the vector-memory counter retires in order) are outstanding; the barrier then publishes the panel to the other waves
  // and at the same time retires panel I-1 in every wave, whose ring slot the DMA of panel I+NBUF-1 overwrites.
  // pieces per issuing wave of the steps T+1 .. T+NBUF-2 (in flight while step T is awaited)
  auto step_in_flight = [](int T) constexpr {
    int n = 0;
    for (int k = 1; k <= NBUF - 2; ++k)
      for (int p = 0; p < PSTEP; ++p) n += SCH::per_issuer((T + k) * PSTEP + p, NW);
    return n;
  };
  auto dma_step = [&](auto TC) {  // all panels of step T
    constexpr int T = decltype(TC)::value;
    tb_static_for<0, PSTEP>([&](auto PC) { dma(std::integral_constant<int, T * PSTEP + decltype(PC)::value>{}); });
  };
  auto step_sync = [&](auto TC) {  // start of step T: its panels have landed and are published; the slot of step T-1 is free
    constexpr int T = decltype(TC)::value;
    static_assert(NBUF == 2 || (32 % NW) == 0, "counted waits need the same number of pieces in every wave");
    constexpr int N = step_in_flight(T);
    TB_ST(4 + 3 * T);
    if (issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    TB_ST(5 + 3 * T);
#ifndef TB_ABL_NOBAR
    __syncthreads();
#endif
    TB_ST(6 + 3 * T);
#ifndef TB_ABL_NODMA
    if (issuer) dma_step(std::integral_constant<int, T + NBUF - 1>{});
#endif
  };
  auto next_panel = [&](auto IC) -> unsigned {
    constexpr int I = decltype(IC)::value;
    if constexpr (I % PSTEP == 0) step_sync(std::integral_constant<int, I / PSTEP>{});
    return pan_lds + ((I / PSTEP) % TB_NBUF) * SLOT_BYTES + (I % PSTEP) * TB_PANEL_BYTES;
  };

  if (issuer) tb_static_for<0, NBUF - 1>([&](auto TC) { dma_step(TC); });
  for (int i = t; i < TC_N; i += NTHREADS) cst[i] = a.consts[i];
  if (MODE && t < 256) ksm[t] = a.ksum[(size_t)b * 256 + t];
  if constexpr (LOADERS > 0) {
    if (wave >= WAVES) {  // a loader wave: the barrier sequence of the computing waves, nothing else
      tb_static_for<0, SCH::NPAN / PSTEP>([&](auto TC) { step_sync(TC); });
      return;
    }
  }

  // ---- X: the input rows, split (mode 0: hidden; mode 1: D)
  half8 xh[8], xl[8];
  float sx;
  {
    const float* src = a.in + (size_t)row * 256;
    float4 va[8], vb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if constexpr (MODE != 0) {
        // the dense layer streams 134 MB of tokens in and out once: non-temporal, so that they do not evict the weight / kv images that
        // 2048 workgroups re-read from the L2 (FETCH_SIZE: 237 MB for 140 MB of compulsory reads with ordinary loads)
        const f32x4 x0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 32 * s + 4 * fg));
        const f32x4 x1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 32 * s + 16 + 4 * fg));
        va[s] = make_float4(x0[0], x0[1], x0[2], x0[3]);
        vb[s] = make_float4(x1[0], x1[1], x1[2], x1[3]);
      } else {
        va[s] = *reinterpret_cast<const float4*>(src + 32 * s + 4 * fg);
        vb[s] = *reinterpret_cast<const float4*>(src + 32 * s + 16 + 4 * fg);
      }
    }
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      m = fmaxf(m, fmaxf(fmaxf(fabsf(va[s].x), fabsf(va[s].y)), fmaxf(fabsf(va[s].z), fabsf(va[s].w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(vb[s].x), fabsf(vb[s].y)), fmaxf(fabsf(vb[s].z), fabsf(vb[s].w))));
    }
    sx = pow2_scale_for(tok_max(m));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float e[8] = {va[s].x, va[s].y, va[s].z, va[s].w, vb[s].x, vb[s].y, vb[s].z, vb[s].w};
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        unsigned hi, lo;
        sam6d_split2_f16(e[u] * sx, e[u + 1] * sx, hi, lo);
        const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
        const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
        xh[s][u] = h2[0];
        xh[s][u + 1] = h2[1];
        xl[s][u] = l2[0];
        xl[s][u + 1] = l2[1];
      }
    }
  }

  // mode 1: the split input rows D stay in registers for the residual add three products later (re-reading them cost a second
  // 134 MB pass over D per layer: 1.7x the layer's algorithmic traffic, rocprof FETCH_SIZE)
  half8 dh[MODE ? 8 : 1], dl[MODE ? 8 : 1];
  const float sd = sx;
  if constexpr (MODE != 0) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      dh[s] = xh[s];
      dl[s] = xl[s];
      tb_opaque(dh[s]);
      tb_opaque(dl[s]);
    }
  }

  TB_ST(1);
  f32x4 acc[16];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  constexpr int P0 = MODE ? 16 : 0;  // first panel of the common part
  if constexpr (MODE != 0) {
    // ---- q = D Wq^T + b, focus, z
    zero_acc();
    tb_static_for<0, 8>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const unsigned p = next_panel(std::integral_constant<int, j>{});
      tb_mma<8, FD>(acc[2 * j], acc[2 * j + 1], p, xh, xl, fr, fg, half);
    });
    {
      const float inv = cst[TC_SC + 0] * (1.0f / sx);
      float n1 = 0.f, n3 = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float4 bq = *reinterpret_cast<const float4*>(cst + TC_BQ + 16 * i + 4 * fg);
        const float4 is = *reinterpret_cast<const float4*>(cst + TC_ISP + 16 * i + 4 * fg);
        const float bb[4] = {bq.x, bq.y, bq.z, bq.w}, ii[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float v = acc[i][u] * inv + bb[u];
          v = ((v > 0.f ? v : 0.f) + 1e-6f) * ii[u];
          n1 += v * v;
          const float c3 = (v * v) * v;
          n3 += c3 * c3;
          acc[i][u] = c3;
        }
      }
      const float f = sqrtf(tok_sum(n1)) / sqrtf(tok_sum(n3));  // phi = c3 / |c3| * |v|
#pragma unroll
      for (int hd = 0; hd < 4; ++hd) {
        float dot = 0.f;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          const int i = 4 * hd + ii;
          const float4 ks = *reinterpret_cast<const float4*>(ksm + 16 * i + 4 * fg);
          acc[i][0] *= f; acc[i][1] *= f; acc[i][2] *= f; acc[i][3] *= f;
          dot += (acc[i][0] * ks.x + acc[i][1] * ks.y) + (acc[i][2] * ks.z + acc[i][3] * ks.w);
        }
        const float z = 1.0f / (tok_sum(dot) + 1e-6f);
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
          acc[4 * hd + ii][0] *= z; acc[4 * hd + ii][1] *= z; acc[4 * hd + ii][2] *= z; acc[4 * hd + ii][3] *= z;
        }
      }
      sx = tb_split_rows<16>(acc, xh, xl);
    }
    // ---- hidden_h = phi(q)_h kv_h  (K = 64 per head: k-steps 2h, 2h+1)
    zero_acc();
    tb_static_for<0, 8>([&](auto J) {
      constexpr int j = decltype(J)::value;
      const unsigned p = next_panel(std::integral_constant<int, 8 + j>{});
      tb_mma<2, FD>(acc[2 * j], acc[2 * j + 1], p, xh + 2 * (j >> 1), xl + 2 * (j >> 1), fr, fg, half);
    });
    {
      const float4 kvi = *reinterpret_cast<const float4*>(a.kvinv + (size_t)b * 4);  // the four heads' image scales
      const float isx = 1.0f / sx;
      const float invh[4] = {kvi.x * isx, kvi.y * isx, kvi.z * isx, kvi.w * isx};
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float inv = invh[i >> 2];  // accumulator tiles 4 h .. 4 h + 3 hold head h's 64 channels
        acc[i][0] *= inv; acc[i][1] *= inv; acc[i][2] *= inv; acc[i][3] *= inv;
      }
      sx = tb_split_rows<16>(acc, xh, xl);
    }
  }

  // ---- y = LayerNorm(hidden Wlin^T + b + residual)
  zero_acc();
  tb_static_for<0, 8>([&](auto J) {
    constexpr int j = decltype(J)::value;
    const unsigned p = next_panel(std::integral_constant<int, P0 + j>{});
    tb_mma<8, FD>(acc[2 * j], acc[2 * j + 1], p, xh, xl, fr, fg, half);
  });
  {
    const float inv = cst[TC_SC + 1] * (1.0f / sx);
    const float* rs = a.resid + (size_t)row * 256;
    const float isd = 1.0f / sd;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      float4 rv;
      if constexpr (MODE != 0) {  // D = (hi + lo) / scale: 22 significand bits of the fp32 input
        rv.x = ((float)dh[i >> 1][4 * (i & 1) + 0] + (float)dl[i >> 1][4 * (i & 1) + 0]) * isd;
        rv.y = ((float)dh[i >> 1][4 * (i & 1) + 1] + (float)dl[i >> 1][4 * (i & 1) + 1]) * isd;
        rv.z = ((float)dh[i >> 1][4 * (i & 1) + 2] + (float)dl[i >> 1][4 * (i & 1) + 2]) * isd;
        rv.w = ((float)dh[i >> 1][4 * (i & 1) + 3] + (float)dl[i >> 1][4 * (i & 1) + 3]) * isd;
      } else {
        rv = *reinterpret_cast<const float4*>(rs + 16 * i + 4 * fg);
      }
      const float4 bl = *reinterpret_cast<const float4*>(cst + TC_BLIN + 16 * i + 4 * fg);
      acc[i][0] = (acc[i][0] * inv + bl.x) + rv.x;
      acc[i][1] = (acc[i][1] * inv + bl.y) + rv.y;
      acc[i][2] = (acc[i][2] * inv + bl.z) + rv.z;
      acc[i][3] = (acc[i][3] * inv + bl.w) + rv.w;
    }
    tb_layernorm(acc, cst + TC_G1, cst + TC_BE1, fg, a.eps);
    sx = tb_split_rows<16>(acc, xh, xl);  // y, kept as hi + lo for the second residual
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      tb_opaque(xh[s]);
      tb_opaque(xl[s]);
    }
  }
  const float sy = sx;

  // ---- out = relu(y Wexp^T + b) Wsq^T, the 512 hidden channels in 4 chunks of 128 that never leave the registers
  zero_acc();
  {
    const float inv_e = cst[TC_SC + 2] * (1.0f / sy), sh = cst[TC_SC + 4];
    tb_static_for<0, 4>([&](auto CC) {
      constexpr int c = decltype(CC)::value;
      half8 hh[4], hl[4];
      tb_static_for<0, 4>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const unsigned p = next_panel(std::integral_constant<int, P0 + 8 + 12 * c + u>{});
        f32x4 ha[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        tb_mma<8, FD>(ha[0], ha[1], p, xh, xl, fr, fg, half);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const float4 be = *reinterpret_cast<const float4*>(cst + TC_BEXP + 128 * c + 32 * u + 16 * w + 4 * fg);
          const float bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
          for (int r = 0; r < 4; r += 2) {
            float v0 = ha[w][r] * inv_e + bb[r], v1 = ha[w][r + 1] * inv_e + bb[r + 1];
            v0 = (v0 > 0.f ? v0 : 0.f) * sh;
            v1 = (v1 > 0.f ? v1 : 0.f) * sh;
            unsigned hi, lo;
            sam6d_split2_f16(v0, v1, hi, lo);
            const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
            const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
            hh[u][4 * w + r] = h2[0];
            hh[u][4 * w + r + 1] = h2[1];
            hl[u][4 * w + r] = l2[0];
            hl[u][4 * w + r + 1] = l2[1];
          }
        }
      });
      tb_static_for<0, 8>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const unsigned p = next_panel(std::integral_constant<int, P0 + 8 + 12 * c + 4 + j>{});
        tb_mma<4, FD>(acc[2 * j], acc[2 * j + 1], p, hh, hl, fr, fg, half);
      });
    });
  }
  {
    const float inv = cst[TC_SC + 3], isy = 1.0f / sy;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 bs = *reinterpret_cast<const float4*>(cst + TC_BSQ + 16 * i + 4 * fg);
      const float bb[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float y = ((float)xh[i >> 1][4 * (i & 1) + r] + (float)xl[i >> 1][4 * (i & 1) + r]) * isy;
        acc[i][r] = (acc[i][r] * inv + bb[r]) + y;
      }
    }
    tb_layernorm(acc, cst + TC_G2, cst + TC_BE2, fg, a.eps);
  }
  TB_ST(2);
  if (valid) {
    float* o = a.out + (size_t)row * 256;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (MODE != 0) __builtin_nontemporal_store(acc[i], reinterpret_cast<f32x4*>(o + 16 * i + 4 * fg));
      else *reinterpret_cast<float4*>(o + 16 * i + 4 * fg) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
    }
  }
  TB_ST(3);
#ifdef TB_STAMP
  if (blockIdx.x < 512)
    for (int i = lane; i < TB_NSTAMP; i += 64) st_base[i] = st_lds[wave][i];
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Measured and removed, round 4: the layer tail with the FFN split over TWO waves per 16-token group.  The primary
// wave of a group ran linear + residual + LayerNorm and handed the normalised row (fp16 hi / lo fragments, 16 KB per group) to a
// secondary wave through LDS; primary and secondary then ran the hidden chunks {0, 1} and {2, 3} at once (a 64 KB ring slot held one
// panel pair of either role, 20 steps instead of 56 panels per chain), and the secondary's partial squeeze accumulators came back
// through LDS for the second residual + LayerNorm: two LDS exchanges per layer (round 3's rejected form split every product and needed
// seven), primary and secondary of a group on one SIMD so that one's row epilogues sit under the other's MFMAs.  232 VGPRs, no spills,
// all 63 tests of tests/test_block_gpu.py green (the hidden chunks summed as (0 + 1) + (2 + 3)).  Times (scratch/run_block.py, one box,
// us per launch): 12608 rows 43.7 against 40.7 for token_block_kernel, 6304 rows 39.7 against 34.0 (the 32-token shape below).
// Why it does not pay: the chain is bound by the LDS, not by the serial epilogues.  Every 16-token wave reads each weight panel from
// LDS whole -- 2 KB of fragments per three MFMAs, i.e. 43 B / clk per wave at the full MFMA rate, 171 B / clk for four waves beside the
// ring's DMA writes, against the ~130-150 B / clk a CU delivers in practice -- so a K = 256 panel takes ~1420 cycles for 768 cycles of
// MFMA issue, and eight computing waves per workgroup double the reads per step while the two roles' panel pairs double the DMA bytes
// in flight: a step of the split kernel took ~3 000 cycles.  Fewer LDS bytes per MFMA would need 32-token waves on v_mfma_f32_32x32x16
// (twice the flops per fragment byte), whose operands + accumulators (128 + 128 registers before the FFN hidden chunks) only fit one
// wave per SIMD and leave half of the SIMDs without a wave at 49 tokens per CU -- priced at 36 us per launch, not built.
// What did help: two computing waves per workgroup where the rows still fill the chip (sam6d_token_block below).
// ---------------------------------------------------------------------------------------------------------------------
// Front of the RPE self-attention layer (RPEMultiHeadAttention.forward, PEM/model/transformer.py:395-405, with proj_p folded into the
// query as attention.hip / rpe.hip describe): one launch instead of three GEMMs
//     qkv = x Wqkv^T + b          (M, 768)   written (q | k | v)
//     qp[n, h, :] = Wp_h^T q_h    (M, 4, 256) the query folded through proj_p, per head (K = 64)
//     qd[n, h, :] = D_c^T qp_h    (M*4, 32)   Chebyshev coefficients of proj_d applied to the folded query
// Same register-chained transposed scheme as token_block_kernel: q stays in the accumulators, becomes the B operand of the proj_p
// fold, whose output becomes the B operand of the D_c product.  Always fp16 x3 (these are the "geometry" folds of matmul mode 2).
// Image: qkv 24 panels (K = 256) | per head 8 panels of Wp_h^T (K = 64) | D_c^T 1 panel (K = 256).
#define RF_IMAGE_BYTES (24 * TB_P256 + 32 * TB_P64 + TB_P256)
struct RfArgs {
  const float* x;            // (M, 256)
  const unsigned char* wimg;
  const float* bias;         // (768)
  float* qkv;                // (M, 768)
  float* qp;                 // (M, 1024)
  float* qd;                 // (M * 4, 32)
  long M;
  float inv_qkv, inv_wp, inv_dc;
  float* vT;                 // optional: (M / n clouds, 256, ldp) the values transposed per cloud (the W operand of the P.v GEMM);
  int n, ldp;                //   then the v third of qkv is not written
};

// VT: the values go to vT (8 scattered 4-byte stores per value panel) instead of the v third of qkv (2 float4 stores).
// Stores are issued for every lane -- rows past M are clamped to row M - 1 and rewrite its values -- so that the number of vector-memory
// operations between two panel DMAs is a compile-time constant: the wait at the head of a panel then lets the previous panel's STORES
// stay in flight (s_waitcnt vmcnt(#stores)) instead of draining them (vmcnt(0) cost a store round trip per panel: 36 per workgroup).
// LOADERS (0 / 4) extra waves issue the panel DMA, as in token_block_kernel; the computing waves then never wait on the vector-memory
// counter at all (their stores stay in flight across panels without the counted waits below).
template <bool VT, int LOADERS = 0>
__global__ __launch_bounds__((4 + LOADERS) * 64, LOADERS ? 1 : 2) void rpe_front_kernel(RfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* pan = lds;  // 2 x TB_PANEL_BYTES
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  const unsigned pan_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)pan;
  const long r0 = (long)blockIdx.x * 64 + wave * 16 + fr;
  const bool valid = r0 < a.M;
  const long row = valid ? r0 : a.M - 1;
  constexpr int NPAN = 24 + 4 * 3;
  // LDS-DMA unit i (32 KiB each): 0..23 the qkv panels (K256); then per head h: two groups of four Wp_h^T panels (K64, 8 KiB each,
  // contiguous in the image: one barrier per group instead of one per panel), then the D_c^T panel (K256)
  auto dma = [&](auto IC) {
    constexpr int I = decltype(IC)::value;
    if constexpr (I < NPAN) {
      constexpr size_t off = I < 24 ? (size_t)I * TB_P256
                                    : (((I - 24) % 3) == 2 ? 24 * (size_t)TB_P256 + 32 * (size_t)TB_P64
                                                           : 24 * (size_t)TB_P256 + (size_t)(((I - 24) / 3) * 8 + ((I - 24) % 3) * 4) * TB_P64);
      constexpr int NP = 32 / 4;
      const unsigned char* src = a.wimg + off;
      unsigned char* dst = pan + (I & 1) * TB_PANEL_BYTES;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int pc = (LOADERS ? wave - 4 : wave) + 4 * k;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                         (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
      }
    }
  };
  // panel I's DMA was issued at the head of panel I - 1; younger than it are only the NST stores of panel I - 1's epilogue
  auto next_panel = [&](auto IC, auto NSTC) -> unsigned {
    constexpr int I = decltype(IC)::value;
    constexpr int NST = decltype(NSTC)::value;
    if constexpr (LOADERS == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NST) : "memory");
    __syncthreads();
    if constexpr (LOADERS == 0) dma(std::integral_constant<int, I + 1>{});
    return pan_lds + (I & 1) * TB_PANEL_BYTES;
  };
  if constexpr (LOADERS > 0) {
    static_assert(LOADERS == 4, "the DMA pieces are dealt to four waves");
    if (wave >= 4) {  // a loader wave: unit I + 1 goes out once unit I has landed and every wave has left unit I - 1
      dma(std::integral_constant<int, 0>{});
      tb_static_for<0, NPAN>([&](auto IC) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        dma(std::integral_constant<int, decltype(IC)::value + 1>{});
      });
      return;
    }
  } else {
    dma(std::integral_constant<int, 0>{});
  }

  half8 xh[8], xl[8];
  float sx;
  {
    const float* src = a.x + (size_t)row * 256;
    float4 va[8], vb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      va[s] = *reinterpret_cast<const float4*>(src + 32 * s + 4 * fg);
      vb[s] = *reinterpret_cast<const float4*>(src + 32 * s + 16 + 4 * fg);
    }
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      m = fmaxf(m, fmaxf(fmaxf(fabsf(va[s].x), fabsf(va[s].y)), fmaxf(fabsf(va[s].z), fabsf(va[s].w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(vb[s].x), fabsf(vb[s].y)), fmaxf(fabsf(vb[s].z), fabsf(vb[s].w))));
    }
    sx = pow2_scale_for(tok_max(m));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float e[8] = {va[s].x, va[s].y, va[s].z, va[s].w, vb[s].x, vb[s].y, vb[s].z, vb[s].w};
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        unsigned hi, lo;
        sam6d_split2_f16(e[u] * sx, e[u + 1] * sx, hi, lo);
        const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
        const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
        xh[s][u] = h2[0];
        xh[s][u + 1] = h2[1];
        xl[s][u] = l2[0];
        xl[s][u + 1] = l2[1];
      }
    }
  }
  // ---- qkv: 24 panels; the q tiles (first 8 panels) are kept for the folds
  f32x4 qa[16];
  const float inv0 = a.inv_qkv * (1.0f / sx);
  float* orow = a.qkv + (size_t)row * 768;
  tb_static_for<0, 24>([&](auto J) {
    constexpr int j = decltype(J)::value;
    // stores of the previous panel: none before panel 0 (the wait also covers the x rows), 8 after a transposed value panel, else 2
    constexpr int prev_st = j == 0 ? 0 : ((VT && j - 1 >= 16) ? 8 : 2);
    const unsigned p = next_panel(std::integral_constant<int, j>{}, std::integral_constant<int, prev_st>{});
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
    tb_mma<8>(c0, c1, p, xh, xl, fr, fg, false);
    const float4 b0 = *reinterpret_cast<const float4*>(a.bias + 32 * j + 4 * fg);
    const float4 b1 = *reinterpret_cast<const float4*>(a.bias + 32 * j + 16 + 4 * fg);
    c0 = f32x4{c0[0] * inv0 + b0.x, c0[1] * inv0 + b0.y, c0[2] * inv0 + b0.z, c0[3] * inv0 + b0.w};
    c1 = f32x4{c1[0] * inv0 + b1.x, c1[1] * inv0 + b1.y, c1[2] * inv0 + b1.z, c1[3] * inv0 + b1.w};
    {
      if constexpr (VT && j >= 16) {  // v channel c of token tok of its cloud -> vT[cloud][c][tok]: 16 consecutive tokens per lane group
        float* vt = a.vT + (size_t)(row / a.n) * 256 * a.ldp + (row % a.n);
        const int cb = 32 * (j - 16) + 4 * fg;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          vt[(size_t)(cb + r) * a.ldp] = c0[r];
          vt[(size_t)(cb + 16 + r) * a.ldp] = c1[r];
        }
      } else {
        *reinterpret_cast<float4*>(orow + 32 * j + 4 * fg) = make_float4(c0[0], c0[1], c0[2], c0[3]);
        *reinterpret_cast<float4*>(orow + 32 * j + 16 + 4 * fg) = make_float4(c1[0], c1[1], c1[2], c1[3]);
      }
    }
    if constexpr (j < 8) {
      qa[2 * j] = c0;
      qa[2 * j + 1] = c1;
    }
  });
  half8 qh[8], ql[8];
  const float sq = tb_split_rows<16>(qa, qh, ql);
  // ---- per head: qp_h = Wp_h^T q_h (8 panels, K = 64 = k-steps 2h, 2h+1 of q), then qd_h = D_c^T qp_h (1 panel, K = 256)
  const float inv1 = a.inv_wp * (1.0f / sq);
  tb_static_for<0, 4>([&](auto HH) {
    constexpr int h = decltype(HH)::value;
    f32x4 pa[16];
    tb_static_for<0, 2>([&](auto GG) {
      constexpr int gq = decltype(GG)::value;
      // younger than this unit's DMA: the last qkv panel's stores (h = 0), the 2 qd stores of the previous head, or nothing (gq = 1)
      constexpr int prev_st = gq == 1 ? 0 : (h == 0 ? (VT ? 8 : 2) : 2);
      const unsigned p4 = next_panel(std::integral_constant<int, 24 + 3 * h + gq>{}, std::integral_constant<int, prev_st>{});
      tb_static_for<0, 4>([&](auto U) {
        constexpr int j = 4 * gq + decltype(U)::value;
        f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
        tb_mma<2>(c0, c1, p4 + decltype(U)::value * TB_P64, qh + 2 * h, ql + 2 * h, fr, fg, false);
        pa[2 * j] = f32x4{c0[0] * inv1, c0[1] * inv1, c0[2] * inv1, c0[3] * inv1};
        pa[2 * j + 1] = f32x4{c1[0] * inv1, c1[1] * inv1, c1[2] * inv1, c1[3] * inv1};
      });
    });
    {
      float* o = a.qp + (size_t)row * 1024 + 256 * h;
#pragma unroll
      for (int i = 0; i < 16; ++i) *reinterpret_cast<float4*>(o + 16 * i + 4 * fg) = make_float4(pa[i][0], pa[i][1], pa[i][2], pa[i][3]);
    }
    half8 ph[8], pl[8];
    const float sp = tb_split_rows<16>(pa, ph, pl);
    const unsigned p = next_panel(std::integral_constant<int, 24 + 3 * h + 2>{}, std::integral_constant<int, 16>{});  // the 16 qp stores
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
    tb_mma<8>(c0, c1, p, ph, pl, fr, fg, false);
    {
      const float inv2 = a.inv_dc * (1.0f / sp);
      float* o = a.qd + ((size_t)row * 4 + h) * 32;
      *reinterpret_cast<float4*>(o + 4 * fg) = make_float4(c0[0] * inv2, c0[1] * inv2, c0[2] * inv2, c0[3] * inv2);
      *reinterpret_cast<float4*>(o + 16 + 4 * fg) = make_float4(c1[0] * inv2, c1[1] * inv2, c1[2] * inv2, c1[3] * inv2);
    }
  });
}

extern "C" long sam6d_rpe_front_image_bytes(void) { return RF_IMAGE_BYTES; }

static int rpe_front_launch(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc,
                            float* qkv, float* qp, float* qd, long M, float* vT, int n, int ldp, void* stream);
extern "C" int sam6d_rpe_front(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc,
                               float* qkv, float* qp, float* qd, long M, void* stream) {
  return rpe_front_launch(x, wimage, bias_qkv, inv_qkv, inv_wp, inv_dc, qkv, qp, qd, M, nullptr, 1, 1, stream);
}
extern "C" int sam6d_rpe_front_vt(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc,
                                  float* qkv, float* qp, float* qd, long M, float* vT, int n, int ldp, void* stream) {
  SAM6D_REQUIRE(vT && n > 0 && ldp >= n && M % n == 0, "rpe_front_vt: vT (M / n, 256, ldp) needs n > 0, ldp >= n, M a multiple of n");
  return rpe_front_launch(x, wimage, bias_qkv, inv_qkv, inv_wp, inv_dc, qkv, qp, qd, M, vT, n, ldp, stream);
}
static int rpe_front_launch(const float* x, const void* wimage, const float* bias_qkv, float inv_qkv, float inv_wp, float inv_dc,
                            float* qkv, float* qp, float* qd, long M, float* vT, int n, int ldp, void* stream) {
  SAM6D_REQUIRE(x && wimage && bias_qkv && qkv && qp && qd && M >= 0, "rpe_front: bad arguments");
  SAM6D_REQUIRE(((((size_t)x) | ((size_t)wimage) | ((size_t)bias_qkv) | ((size_t)qkv) | ((size_t)qp) | ((size_t)qd)) & 15) == 0,
                "rpe_front: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  static unsigned long long done = 0;
  if (sam6d_first_use_on_device(&done)) {
    hipError_t e = hipFuncSetAttribute((const void*)rpe_front_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_PANEL_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rpe_front_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_PANEL_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rpe_front_kernel<false, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_PANEL_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)rpe_front_kernel<true, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_PANEL_BYTES);
    if (e != hipSuccess) {
      sam6d_set_error("rpe_front: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&done);
  }
  RfArgs a{x, (const unsigned char*)wimage, bias_qkv, qkv, qp, qd, M, inv_qkv, inv_wp, inv_dc, vT, n, ldp};
  static int loaders = -1;  // four loader waves beside the four computing ones (SAM6D_FRONT_LOADERS=0: the plain shape, for A/B runs)
  if (loaders < 0) {
    const char* e = getenv("SAM6D_FRONT_LOADERS");
    loaders = (e && e[0] == '0') ? 0 : 4;
  }
  const dim3 g((unsigned)((M + 63) / 64));
  if (vT && loaders)
    hipLaunchKernelGGL((rpe_front_kernel<true, 4>), g, dim3(512), 2 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  else if (vT)
    hipLaunchKernelGGL((rpe_front_kernel<true, 0>), g, dim3(256), 2 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  else if (loaders)
    hipLaunchKernelGGL((rpe_front_kernel<false, 4>), g, dim3(512), 2 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((rpe_front_kernel<false, 0>), g, dim3(256), 2 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("rpe_front");
}

// ---------------------------------------------------------------------------------------------------------------------
// out_proj + F.normalize + the operand split of the fine similarity in one pass over the dense tokens
// (FinePointMatching.forward: out_proj, PEM/model/fine_point_matching.py:70-72; compute_feature_similarity: F.normalize(dim=2),
// PEM/utils/model_utils.py:141-142):   y = x W^T + b;   fh | fl = fp16 hi / lo of (y / max(|y|, 1e-12)) * 2^10
// -- what sam6d_gemm_nt followed by finematch.hip's fm_prep_kernel produce (a 134 MB fp32 intermediate written and read back at
// B = 32).  One workgroup = 64 token rows on four computing waves + four loader waves (8 panels, two per ring slot), the same
// register-chained transposed product as token_block_kernel; rows are read non-temporally (each exactly once).
struct OsArgs {
  const float* x;            // (M, 256)
  const unsigned char* wimg; // sam6d_pack_panels(W, 256 rows, k0 = 0, ksteps = 8): 8 panels
  const float* bias;         // (256)
  _Float16* fh;              // (M, 256)
  _Float16* fl;
  long M;
  float inv_w;               // 1 / pack scale of W
  int half;
};

// LOADERS = 4, PSTEP = 2: four loader waves, two panels per ring slot (one workgroup per CU: 128 KB of LDS); LOADERS = 0, PSTEP = 1:
// the computing waves issue their own DMA, one panel per slot (64 KB: two workgroups per CU, one's row loads / stores beside the
// other's products) -- the shape for this pure streaming pass.
template <int LOADERS, int PSTEP>
__global__ __launch_bounds__((4 + LOADERS) * 64, LOADERS ? 1 : 2) void out_split_kernel(OsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  const unsigned pan_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  constexpr int SLOT = PSTEP * TB_PANEL_BYTES, NSTEP = 8 / PSTEP;
  auto dma_step = [&](int T) {  // step T = PSTEP panels into slot T & 1, dealt to the four issuing waves
    const unsigned char* src = a.wimg + (size_t)T * SLOT;
    unsigned char* dst = lds + (T & 1) * SLOT;
#pragma unroll
    for (int k = 0; k < 8 * PSTEP; ++k) {
      const int pc = (LOADERS ? wave - 4 : wave) + 4 * k;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                       (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  if constexpr (LOADERS > 0) {
    if (wave >= 4) {
      dma_step(0);
#pragma unroll
      for (int T = 0; T < NSTEP; ++T) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (T + 1 < NSTEP) dma_step(T + 1);
      }
      return;
    }
  } else {
    dma_step(0);
  }
  const long r0 = (long)blockIdx.x * 64 + wave * 16 + fr;
  const bool valid = r0 < a.M;
  const long row = valid ? r0 : a.M - 1;
  half8 xh[8], xl[8];
  float sx;
  {
    const float* src = a.x + (size_t)row * 256;
    f32x4 va[8], vb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      va[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 32 * s + 4 * fg));
      vb[s] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(src + 32 * s + 16 + 4 * fg));
    }
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) m = fmaxf(m, fmaxf(fabsf(va[s][r]), fabsf(vb[s][r])));
    sx = pow2_scale_for(tok_max(m));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float e[8] = {va[s][0], va[s][1], va[s][2], va[s][3], vb[s][0], vb[s][1], vb[s][2], vb[s][3]};
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        unsigned hi, lo;
        sam6d_split2_f16(e[u] * sx, e[u + 1] * sx, hi, lo);
        const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
        const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
        xh[s][u] = h2[0];
        xh[s][u + 1] = h2[1];
        xl[s][u] = l2[0];
        xl[s][u + 1] = l2[1];
      }
    }
  }
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool half = a.half != 0;
  tb_static_for<0, 8>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr (j % PSTEP == 0) {  // step j / PSTEP has landed and is published; the slot of the step before is free
      if constexpr (LOADERS == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if constexpr (LOADERS == 0 && j / PSTEP + 1 < NSTEP) dma_step(j / PSTEP + 1);
    }
    const unsigned p = pan_lds + ((j / PSTEP) & 1) * SLOT + (j % PSTEP) * TB_PANEL_BYTES;
    tb_mma<8>(acc[2 * j], acc[2 * j + 1], p, xh, xl, fr, fg, half);
  });
  const float inv = a.inv_w * (1.0f / sx);
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float4 bb = *reinterpret_cast<const float4*>(a.bias + 16 * i + 4 * fg);
    acc[i][0] = acc[i][0] * inv + bb.x;
    acc[i][1] = acc[i][1] * inv + bb.y;
    acc[i][2] = acc[i][2] * inv + bb.z;
    acc[i][3] = acc[i][3] * inv + bb.w;
    ss += (acc[i][0] * acc[i][0] + acc[i][1] * acc[i][1]) + (acc[i][2] * acc[i][2] + acc[i][3] * acc[i][3]);
  }
  const float d = fmaxf(sqrtf(tok_sum(ss)), 1e-12f);
  if (valid) {
    _Float16* oh = a.fh + (size_t)row * 256 + 4 * fg;
    _Float16* ol = a.fl + (size_t)row * 256 + 4 * fg;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      unsigned h01, l01, h23, l23;
      sam6d_split2_f16((acc[i][0] / d) * 1024.0f, (acc[i][1] / d) * 1024.0f, h01, l01);
      sam6d_split2_f16((acc[i][2] / d) * 1024.0f, (acc[i][3] / d) * 1024.0f, h23, l23);
      *reinterpret_cast<tb_u2*>(oh + 16 * i) = tb_u2{h01, h23};
      *reinterpret_cast<tb_u2*>(ol + 16 * i) = tb_u2{l01, l23};
    }
  }
}

// (Measured and removed, round 4: the same pass as a PERSISTENT kernel -- one workgroup per CU, four computing + four loader waves, the
// loader waves streaming the eight panels round and round without a bubble at the tile seams, each computing wave requesting the NEXT
// tile's rows right after splitting the current ones (no LDS-DMA in the computing waves' vector-memory queue, so nothing orders the row
// loads behind a panel wait).  Same bits; 117.5 / 117.9 us against 111.8 us per launch at M = 131 136 (scratch/ub_stream.py).  The row
// latency was not the bound: every 64-row tile re-streams the 256 KB weight image through the LDS-DMA path, 2048 tiles x 256 KB = 537 MB
// per launch at ~18 GB/s per CU (4.6 TB/s chip-wide) -- the pass moves twice as many weight bytes L2 -> LDS as token bytes HBM -> registers.
// Fewer weight bytes per row need wider row tiles (128 rows on eight computing waves without loader waves) or resident weights (256 KB
// of fp16 hi / lo do not fit the 160 KB of LDS; half the output channels do, but the row norm needs all 256).)
extern "C" int sam6d_linear_norm_split(const float* x, const void* wimage, const float* bias, float inv_w_scale, void* fh, void* fl,
                                       long M, void* stream) {
  SAM6D_REQUIRE(x && wimage && bias && fh && fl && M >= 0 && inv_w_scale > 0.f, "linear_norm_split: bad arguments");
  SAM6D_REQUIRE(((((size_t)x) | ((size_t)wimage) | ((size_t)bias) | ((size_t)fh) | ((size_t)fl)) & 15) == 0,
                "linear_norm_split: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  static unsigned long long done = 0;
  if (sam6d_first_use_on_device(&done)) {
    hipError_t e = hipFuncSetAttribute((const void*)out_split_kernel<4, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TB_PANEL_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)out_split_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TB_PANEL_BYTES);
    if (e != hipSuccess) {
      sam6d_set_error("linear_norm_split: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&done);
  }
  OsArgs a{x, (const unsigned char*)wimage, bias, (_Float16*)fh, (_Float16*)fl, M, inv_w_scale, sam6d_half_for(1)};
  static int shape = -1;  // SAM6D_OUT_SPLIT_LOADERS=1: the one-workgroup-per-CU shape with loader waves (A/B runs)
  if (shape < 0) {
    const char* e = getenv("SAM6D_OUT_SPLIT_LOADERS");
    shape = (e && e[0] == '1') ? 1 : 0;
  }
  const dim3 g((unsigned)((M + 63) / 64));
  if (shape)
    hipLaunchKernelGGL((out_split_kernel<4, 2>), g, dim3(512), 4 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((out_split_kernel<0, 1>), g, dim3(256), 2 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("linear_norm_split");
}

// ---------------------------------------------------------------------------------------------------------------------
// y = x W^T + b for a few thousand 256-channel token rows (the sparse-token projections: in_proj / out_proj of the coarse stage,
// [proj_k; proj_v] of the dense layers' memory tokens -- PEM/model/coarse_point_matching.py:35-38, 61; PEM/model/transformer.py:556-558):
// 197 workgroups of 64 rows on the panel machinery of token_block_kernel (four computing + four loader waves, two panels per ring
// slot) instead of the generic GEMM's 64 x 64 tiles, whose ~1600 single-k-chunk workgroups are launch-latency-bound at this size.
// Rows may be strided per cloud on both sides (row R = cloud R / rpb, token R % rpb): the bg slot of a token buffer is skipped in place.
struct RlArgs {
  const float* x;
  const unsigned char* wimg;  // sam6d_pack_panels(W, 32 NP rows, k0 = 0, ksteps = 8): NP panels
  const float* bias;          // (32 NP) or null
  float* out;
  long M;
  int rpb;                    // rows per cloud
  long x_bs, x_r0, o_bs, o_r0;  // input / output: rows between two clouds, first row of a cloud
  float inv_w;
  int half;
};

template <int NP>
__global__ __launch_bounds__(512, 1) void rows_linear_kernel(RlArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  const unsigned pan_lds = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  constexpr int SLOT = 2 * TB_PANEL_BYTES, NSTEP = NP / 2;
  static_assert(NP % 2 == 0, "two panels per step");
  if (wave >= 4) {  // loader waves
    auto dma_step = [&](int T) {
      const unsigned char* src = a.wimg + (size_t)T * SLOT;
      unsigned char* dst = lds + (T & 1) * SLOT;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int pc = (wave - 4) + 4 * k;
        __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                         (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
      }
    };
    dma_step(0);
#pragma unroll
    for (int T = 0; T < NSTEP; ++T) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (T + 1 < NSTEP) dma_step(T + 1);
    }
    return;
  }
  const long r0 = (long)blockIdx.x * 64 + wave * 16 + fr;
  const bool valid = r0 < a.M;
  const long R = valid ? r0 : a.M - 1;
  const long cb = R / a.rpb, cr = R - cb * a.rpb;
  half8 xh[8], xl[8];
  float sx;
  {
    const float* src = a.x + (size_t)(cb * a.x_bs + a.x_r0 + cr) * 256;
    float4 va[8], vb[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      va[s] = *reinterpret_cast<const float4*>(src + 32 * s + 4 * fg);
      vb[s] = *reinterpret_cast<const float4*>(src + 32 * s + 16 + 4 * fg);
    }
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      m = fmaxf(m, fmaxf(fmaxf(fabsf(va[s].x), fabsf(va[s].y)), fmaxf(fabsf(va[s].z), fabsf(va[s].w))));
      m = fmaxf(m, fmaxf(fmaxf(fabsf(vb[s].x), fabsf(vb[s].y)), fmaxf(fabsf(vb[s].z), fabsf(vb[s].w))));
    }
    sx = pow2_scale_for(tok_max(m));
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const float e[8] = {va[s].x, va[s].y, va[s].z, va[s].w, vb[s].x, vb[s].y, vb[s].z, vb[s].w};
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        unsigned hi, lo;
        sam6d_split2_f16(e[u] * sx, e[u + 1] * sx, hi, lo);
        const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
        const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
        xh[s][u] = h2[0];
        xh[s][u + 1] = h2[1];
        xl[s][u] = l2[0];
        xl[s][u + 1] = l2[1];
      }
    }
  }
  const bool half = a.half != 0;
  const float inv = a.inv_w * (1.0f / sx);
  float* orow = a.out + (size_t)(cb * a.o_bs + a.o_r0 + cr) * (32 * NP);
  tb_static_for<0, NP>([&](auto J) {
    constexpr int j = decltype(J)::value;
    if constexpr ((j & 1) == 0) __syncthreads();  // step j / 2 has landed (loader waves) and is published
    const unsigned p = pan_lds + ((j >> 1) & 1) * SLOT + (j & 1) * TB_PANEL_BYTES;
    f32x4 c0 = f32x4{0.f, 0.f, 0.f, 0.f}, c1 = f32x4{0.f, 0.f, 0.f, 0.f};
    tb_mma<8>(c0, c1, p, xh, xl, fr, fg, half);
    float4 b0 = make_float4(0.f, 0.f, 0.f, 0.f), b1 = b0;
    if (a.bias) {
      b0 = *reinterpret_cast<const float4*>(a.bias + 32 * j + 4 * fg);
      b1 = *reinterpret_cast<const float4*>(a.bias + 32 * j + 16 + 4 * fg);
    }
    if (valid) {
      *reinterpret_cast<float4*>(orow + 32 * j + 4 * fg) = make_float4(c0[0] * inv + b0.x, c0[1] * inv + b0.y, c0[2] * inv + b0.z, c0[3] * inv + b0.w);
      *reinterpret_cast<float4*>(orow + 32 * j + 16 + 4 * fg) = make_float4(c1[0] * inv + b1.x, c1[1] * inv + b1.y, c1[2] * inv + b1.z, c1[3] * inv + b1.w);
    }
  });
}

extern "C" int sam6d_rows_linear(const float* x, const void* wimage, int npanels, const float* bias, float inv_w_scale, float* out,
                                 long M, int rows_per_cloud, long x_cloud_rows, long x_row0, long out_cloud_rows, long out_row0,
                                 void* stream) {
  SAM6D_REQUIRE(x && wimage && out && M >= 0 && inv_w_scale > 0.f, "rows_linear: bad arguments");
  SAM6D_REQUIRE(npanels == 8 || npanels == 16, "rows_linear: built for 256 or 512 output channels (8 or 16 panels), got %d panels", npanels);
  SAM6D_REQUIRE(rows_per_cloud > 0 && x_cloud_rows >= rows_per_cloud && out_cloud_rows >= rows_per_cloud && x_row0 >= 0 && out_row0 >= 0,
                "rows_linear: bad row mapping");
  SAM6D_REQUIRE(((((size_t)x) | ((size_t)wimage) | ((size_t)bias) | ((size_t)out)) & 15) == 0, "rows_linear: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  static unsigned long long done = 0;
  if (sam6d_first_use_on_device(&done)) {
    hipError_t e = hipFuncSetAttribute((const void*)rows_linear_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TB_PANEL_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)rows_linear_kernel<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * TB_PANEL_BYTES);
    if (e != hipSuccess) {
      sam6d_set_error("rows_linear: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&done);
  }
  RlArgs a{x, (const unsigned char*)wimage, bias, out, M, rows_per_cloud, x_cloud_rows, x_row0, out_cloud_rows, out_row0, inv_w_scale,
           sam6d_half_for(1)};
  const dim3 g((unsigned)((M + 63) / 64));
  if (npanels == 8)
    hipLaunchKernelGGL(rows_linear_kernel<8>, g, dim3(512), 4 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(rows_linear_kernel<16>, g, dim3(512), 4 * TB_PANEL_BYTES, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("rows_linear");
}

#define TB_LDS_BYTES(NBUF) ((NBUF) * TB_PANEL_BYTES + (TC_N + 256) * 4)
#define TB_LDS_BYTES2(NBUF, PSTEP) ((NBUF) * (PSTEP) * TB_PANEL_BYTES + (TC_N + 256) * 4)

// Two shapes of the same kernel: 4 waves x 16 tokens with a 2-slot panel ring (77 KB of LDS: two workgroups per CU, which run out
// of step, so one's row epilogues overlap the other's MFMAs), and 8 waves x 16 tokens with a 4-slot ring (one workgroup per CU).
// SAM6D_BLOCK_SHAPE=8 selects the latter (kept for A/B measurements).
// The 197-token layers (sam6d_token_block) default to 4 computing + 4 loader waves with two panels per ring slot (shape 142); "4" is the
// plain 4-wave shape there, "4L" / "4P" loader waves / two-panel steps alone.
static int tb_shape() {
  static int shape = 0;
  if (!shape) {
    const char* e = getenv("SAM6D_BLOCK_SHAPE");
    shape = !e ? 142 : e[0] == '8' ? 8 : (e[0] == '4' && e[1] == '4') ? 44 : (e[0] == '4' && e[1] == 'L') ? 141 :
            (e[0] == '4' && e[1] == 'P') ? 42 : e[0] == '4' ? 4 : 142;
  }
  return shape;
}

template <class K>
static int tb_attr(K kernel, int bytes) {
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    sam6d_set_error("token_block: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  return 0;
}

static int tb_set_attr() {
  static unsigned long long done0 = 0;
  if (sam6d_first_use_on_device(&done0)) {
    int rc = tb_attr(token_block_kernel<0, 4, 2>, TB_LDS_BYTES(2));
    if (!rc) rc = tb_attr(token_block_kernel<1, 4, 2>, TB_LDS_BYTES(2));
    if (!rc) rc = tb_attr(token_block_kernel<0, 8, 4>, TB_LDS_BYTES(4));
    if (!rc) rc = tb_attr(token_block_kernel<0, 4, 4>, TB_LDS_BYTES(4));
    if (!rc) rc = tb_attr(token_block_kernel<1, 8, 4>, TB_LDS_BYTES(4));
    if (!rc) rc = tb_attr(token_block_kernel<0, 4, 2, TB_FD, 4, 2>, TB_LDS_BYTES2(2, 2));
    if (!rc) rc = tb_attr(token_block_kernel<0, 4, 2, TB_FD, 4, 1>, TB_LDS_BYTES2(2, 1));
    if (!rc) rc = tb_attr(token_block_kernel<0, 4, 2, TB_FD, 0, 2>, TB_LDS_BYTES2(2, 2));
    if (!rc) rc = tb_attr(token_block_kernel<0, 2, 2, TB_FD, 4, 2>, TB_LDS_BYTES2(2, 2));
    if (!rc) rc = tb_attr(token_block_kernel<0, 2, 2, TB_FD, 2, 2>, TB_LDS_BYTES2(2, 2));
    if (rc) return rc;
    sam6d_setup_done_on_device(&done0);
  }
  return 0;
}

extern "C" int sam6d_token_block(const float* hidden, const float* x, const void* wimage, const float* consts, float* out, long M,
                                 float eps, void* stream) {
  SAM6D_REQUIRE(hidden && x && wimage && consts && out && M >= 0, "token_block: bad arguments");
  SAM6D_REQUIRE(((((size_t)hidden) | ((size_t)x) | ((size_t)out) | ((size_t)wimage)) & 15) == 0, "token_block: pointers must be 16-byte aligned");
  if (M == 0) return 0;
  int rc = tb_set_attr();
  if (rc) return rc;
  TbArgs a{hidden, x, out, (const unsigned char*)wimage, consts, nullptr, nullptr, nullptr, M, 0, 0, 0, eps,
           sam6d_half_for(1)};
  const dim3 g64((unsigned)((M + 63) / 64));
  // Round 4: 32-token workgroups (two computing waves) while they all fit the chip at once.  A wave's chain -- 56 weight panels through
  // the LDS ring -- is what a launch takes whatever the number of workgroups; the 6304-row launches (the cross layers: 32 clouds x 197)
  // filled 99 of the 256 CUs with four computing waves each, which contend for the LDS reads of every panel (4 x 32 KB per panel) and
  // leave 157 CUs idle; with two computing waves per workgroup the same rows use 197 CUs and a panel is read twice, not four times.
  // A token's arithmetic does not depend on the workgroup shape (everything is per 16-token wave), so the results are bit-identical.
  static int narrow = -1;
  if (narrow < 0) {
    const char* e = getenv("SAM6D_TB_NARROW");  // A/B: 0 = always 64-token workgroups; 2 = two loader waves instead of four
    narrow = e ? atoi(e) : 1;
  }
  int dev_ = 0, cus = 256;
  if (hipGetDevice(&dev_) == hipSuccess) {
    static int cu_cache[SAM6D_MAX_DEVICES];
    if (dev_ >= 0 && dev_ < SAM6D_MAX_DEVICES) {
      if (!cu_cache[dev_] && hipDeviceGetAttribute(&cu_cache[dev_], hipDeviceAttributeMultiprocessorCount, dev_) != hipSuccess) cu_cache[dev_] = 256;
      cus = cu_cache[dev_];
    }
  }
  if (narrow && tb_shape() == 142 && (M + 31) / 32 <= cus) {
    const dim3 g32((unsigned)((M + 31) / 32));
    if (narrow == 2)
      hipLaunchKernelGGL((token_block_kernel<0, 2, 2, TB_FD, 2, 2>), g32, dim3(256), TB_LDS_BYTES2(2, 2), (hipStream_t)stream, a);
    else
      hipLaunchKernelGGL((token_block_kernel<0, 2, 2, TB_FD, 4, 2>), g32, dim3(384), TB_LDS_BYTES2(2, 2), (hipStream_t)stream, a);
  } else if (tb_shape() == 142)
    hipLaunchKernelGGL((token_block_kernel<0, 4, 2, TB_FD, 4, 2>), g64, dim3(512), TB_LDS_BYTES2(2, 2), (hipStream_t)stream, a);
  else if (tb_shape() == 141)
    hipLaunchKernelGGL((token_block_kernel<0, 4, 2, TB_FD, 4, 1>), g64, dim3(512), TB_LDS_BYTES2(2, 1), (hipStream_t)stream, a);
  else if (tb_shape() == 42)
    hipLaunchKernelGGL((token_block_kernel<0, 4, 2, TB_FD, 0, 2>), g64, dim3(256), TB_LDS_BYTES2(2, 2), (hipStream_t)stream, a);
  else if (tb_shape() == 44)
    hipLaunchKernelGGL((token_block_kernel<0, 4, 4>), dim3((unsigned)((M + 63) / 64)), dim3(256), TB_LDS_BYTES(4), (hipStream_t)stream, a);
  else if (tb_shape() == 8)
    hipLaunchKernelGGL((token_block_kernel<0, 8, 4>), dim3((unsigned)((M + 127) / 128)), dim3(512), TB_LDS_BYTES(4), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((token_block_kernel<0, 4, 2>), dim3((unsigned)((M + 63) / 64)), dim3(256), TB_LDS_BYTES(2), (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("token_block");
}

extern "C" int sam6d_linattn_layer(const float* D, const void* wimage, const float* consts, const void* kvimage, const float* kvinv,
                                   const float* ksum, float* Dout, int B, int I, int row0, float eps, void* stream) {
  SAM6D_REQUIRE(D && wimage && consts && kvimage && kvinv && ksum && Dout && B >= 0 && I > 0 && row0 >= 0 && row0 < I,
                "linattn_layer: bad arguments");
  SAM6D_REQUIRE(((((size_t)D) | ((size_t)Dout) | ((size_t)wimage) | ((size_t)kvimage)) & 15) == 0, "linattn_layer: pointers must be 16-byte aligned");
  if (B == 0) return 0;
  int rc = tb_set_attr();
  if (rc) return rc;
  const int dshape = tb_shape() == 8 ? 8 : 4;  // (128-token workgroups for the dense layer alone: 0.342 ms against 0.341 ms)
  const int tok = dshape == 8 ? 128 : 64;  // tokens per workgroup of the kernel shape launched below
  const int tiles = (I - row0 + tok - 1) / tok;
  SAM6D_REQUIRE((long)B * tiles < 2147483647L, "linattn_layer: too many tiles");
  TbArgs a{D, nullptr, Dout, (const unsigned char*)wimage, consts, (const unsigned char*)kvimage, kvinv, ksum, 0, I, row0, tiles, eps,
           sam6d_half_for(1)};
  if (dshape == 8)
    hipLaunchKernelGGL((token_block_kernel<1, 8, 4>), dim3((unsigned)(B * tiles)), dim3(512), TB_LDS_BYTES(4), (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL((token_block_kernel<1, 4, 2>), dim3((unsigned)(B * tiles)), dim3(256), TB_LDS_BYTES(2), (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("linattn_layer");
}
