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

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define TB_TOK 128                              // tokens per workgroup (8 waves x 16)
#define TB_ROWB(KS) ((KS) * 128 + 32)           // bytes of one weight row in a panel: KS*32 hi halves | KS*32 lo halves | 32 B pad
#define TB_PIECES(KS) ((32 * TB_ROWB(KS) + 1023) / 1024)  // 1 KiB DMA pieces per 32-row panel: 33 (K=256), 17 (K=128), 9 (K=64)
#define TB_PANEL_BYTES (TB_PIECES(8) * 1024)    // 33 792
#define TB_P256 (TB_PIECES(8) * 1024)
#define TB_P128 (TB_PIECES(4) * 1024)
#define TB_P64 (TB_PIECES(2) * 1024)
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
  const int rowb = TB_ROWB(KS), pbytes = TB_PIECES(KS) * 1024;
  unsigned char* out = dst + (size_t)panel * pbytes;
  for (int i = threadIdx.x; i < pbytes / 2; i += 256) reinterpret_cast<_Float16*>(out)[i] = (_Float16)0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < 32 * KS * 32; i += 256) {
    const int m = i / (KS * 32), p = i % (KS * 32);
    const int col = k0 + 32 * (p >> 5) + tb_slot_channel(p & 31);
    const float v = W[(size_t)(panel * 32 + m) * ldw + col] * scale;
    const _Float16 hi = (_Float16)v;
    _Float16* row = reinterpret_cast<_Float16*>(out + (size_t)m * rowb);
    row[p] = hi;
    row[KS * 32 + p] = (_Float16)(v - (float)hi);
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

// per cloud: kv^T (4 heads x 64 d x 64 c) -> 8 panels of 32 d-rows x K = 64 (head h: panels 2h, 2h+1), scaled by a power of two
// chosen from the cloud's max |kv|; inv[b] = 1 / scale.
__global__ __launch_bounds__(256) void tb_kv_pack_kernel(const float* __restrict__ kvT, unsigned char* __restrict__ dst,
                                                         float* __restrict__ inv) {
  __shared__ float red[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const float* src = kvT + (size_t)b * 16384;
  float m = 0.f;
  for (int i = t; i < 16384; i += 256) m = fmaxf(m, fabsf(src[i]));
  m = wave_max_dpp(m);
  if ((t & 63) == 0) red[t >> 6] = m;
  __syncthreads();
  const float scale = pow2_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
  if (t == 0) inv[b] = 1.0f / scale;
  unsigned char* out = dst + (size_t)b * (8 * TB_P64);
  for (int i = t; i < 8 * TB_P64 / 2; i += 256) reinterpret_cast<_Float16*>(out)[i] = (_Float16)0.f;
  __syncthreads();
  for (int i = t; i < 16384; i += 256) {
    const int hd = i >> 12, d = (i >> 6) & 63, p = i & 63;  // p: slot inside the 64-wide K
    const int c = 32 * (p >> 5) + tb_slot_channel(p & 31);
    const float v = src[(hd * 64 + d) * 64 + c] * scale;
    const _Float16 hi = (_Float16)v;
    _Float16* row = reinterpret_cast<_Float16*>(out + (size_t)(2 * hd + (d >> 5)) * TB_P64 + (size_t)(d & 31) * TB_ROWB(2));
    row[p] = hi;
    row[64 + p] = (_Float16)(v - (float)hi);
  }
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
// one 32-row panel = two 16-row out tiles; KS k-steps of 32
template <int KS>
__device__ __forceinline__ void tb_mma(f32x4& acc0, f32x4& acc1, const unsigned char* __restrict__ panel, const half8* __restrict__ xh,
                                       const half8* __restrict__ xl, int fr, int fg) {
  const unsigned char* row0 = panel + fr * TB_ROWB(KS) + fg * 16;
  const unsigned char* row1 = row0 + 16 * TB_ROWB(KS);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8 ah0 = *reinterpret_cast<const half8*>(row0 + s * 64);
    const half8 al0 = *reinterpret_cast<const half8*>(row0 + KS * 64 + s * 64);
    const half8 ah1 = *reinterpret_cast<const half8*>(row1 + s * 64);
    const half8 al1 = *reinterpret_cast<const half8*>(row1 + KS * 64 + s * 64);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, xh[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, xh[s], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xl[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xl[s], acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, xh[s], acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, xh[s], acc1, 0, 0, 0);
  }
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
typedef unsigned tb_u32x4 __attribute__((ext_vector_type(4)));
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
    for (int r = 0; r < 4; ++r) {
      const float x = v[t][r] * sc;
      const _Float16 hi = (_Float16)x;
      xh[t >> 1][4 * (t & 1) + r] = hi;
      xl[t >> 1][4 * (t & 1) + r] = (_Float16)(x - (float)hi);
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
  const float* kvinv;          // mode 1: (B)
  const float* ksum;           // mode 1: (B, 256)
  long M;                 // mode 0: rows
  int I, row0, tiles_per_b;  // mode 1: rows per cloud, first row handled, tiles per cloud
  float eps;
};

template <int MODE>
__global__ __launch_bounds__(512) void token_block_kernel(TbArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* pan = lds;                                                   // 2 x TB_PANEL_BYTES
  float* cst = reinterpret_cast<float*>(lds + 2 * TB_PANEL_BYTES);            // TC_N floats
  float* ksm = cst + TC_N;                                                    // 256 floats (mode 1)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, fr = lane & 15, fg = lane >> 4;
  constexpr int NPAN = (MODE ? 16 : 0) + 8 + 4 * 12;

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
  auto dma = [&](int i) {
    const unsigned char* src = a.wimg;
    int pieces = TB_PIECES(8);
    int k = i;
    bool done = false;
    if (MODE) {
      if (k < 8) { src = a.wimg + TB_Q_OFF + (size_t)k * TB_P256; done = true; }
      else if (k < 16) { src = kvp + (size_t)(k - 8) * TB_P64; pieces = TB_PIECES(2); done = true; }
      k -= 16;
    }
    if (!done) {
      if (k < 8) src = a.wimg + (size_t)k * TB_P256;
      else {
        k -= 8;
        const int c = k / 12, u = k % 12;
        const unsigned char* base = a.wimg + 8 * TB_P256 + (size_t)c * TB_CHUNK_BYTES;
        if (u < 4) src = base + (size_t)u * TB_P256;
        else { src = base + 4 * TB_P256 + (size_t)(u - 4) * TB_P128; pieces = TB_PIECES(4); }
      }
    }
    unsigned char* dst = pan + (i & 1) * TB_PANEL_BYTES;
    for (int pc = wave; pc < pieces; pc += 8)
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                       (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
  };
  int pi = 0;
  // the barrier both publishes panel pi (every wave's DMA pieces have landed: vmcnt) and retires panel pi-1, whose buffer the
  // next DMA overwrites
  auto next_panel = [&]() -> const unsigned char* {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (pi + 1 < NPAN) dma(pi + 1);
    const unsigned char* p = pan + (pi & 1) * TB_PANEL_BYTES;
    ++pi;
    return p;
  };

  dma(0);
  for (int i = t; i < TC_N; i += 512) cst[i] = a.consts[i];
  if (MODE && t < 256) ksm[t] = a.ksum[(size_t)b * 256 + t];

  // ---- X: the input rows, split (mode 0: hidden; mode 1: D)
  half8 xh[8], xl[8];
  float sx;
  {
    const float* src = a.in + (size_t)row * 256;
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
      for (int u = 0; u < 8; ++u) {
        const float x = e[u] * sx;
        const _Float16 hi = (_Float16)x;
        xh[s][u] = hi;
        xl[s][u] = (_Float16)(x - (float)hi);
      }
    }
  }

  f32x4 acc[16];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  if (MODE) {
    // ---- q = D Wq^T + b, focus, z
    zero_acc();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned char* p = next_panel();
      tb_mma<8>(acc[2 * j], acc[2 * j + 1], p, xh, xl, fr, fg);
    }
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
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const unsigned char* p = next_panel();
      tb_mma<2>(acc[2 * j], acc[2 * j + 1], p, xh + 2 * (j >> 1), xl + 2 * (j >> 1), fr, fg);
    }
    {
      const float inv = a.kvinv[b] * (1.0f / sx);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i][0] *= inv; acc[i][1] *= inv; acc[i][2] *= inv; acc[i][3] *= inv;
      }
      sx = tb_split_rows<16>(acc, xh, xl);
    }
  }

  // ---- y = LayerNorm(hidden Wlin^T + b + residual)
  zero_acc();
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned char* p = next_panel();
    tb_mma<8>(acc[2 * j], acc[2 * j + 1], p, xh, xl, fr, fg);
  }
  {
    const float inv = cst[TC_SC + 1] * (1.0f / sx);
    const float* rs = (MODE ? a.in : a.resid) + (size_t)row * 256;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float4 rv = *reinterpret_cast<const float4*>(rs + 16 * i + 4 * fg);
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
#pragma unroll 1
    for (int c = 0; c < 4; ++c) {
      half8 hh[4], hl[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned char* p = next_panel();
        f32x4 ha[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        tb_mma<8>(ha[0], ha[1], p, xh, xl, fr, fg);
#pragma unroll
        for (int w = 0; w < 2; ++w) {
          const float4 be = *reinterpret_cast<const float4*>(cst + TC_BEXP + 128 * c + 32 * u + 16 * w + 4 * fg);
          const float bb[4] = {be.x, be.y, be.z, be.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float v = ha[w][r] * inv_e + bb[r];
            v = (v > 0.f ? v : 0.f) * sh;
            const _Float16 hi = (_Float16)v;
            hh[u][4 * w + r] = hi;
            hl[u][4 * w + r] = (_Float16)(v - (float)hi);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned char* p = next_panel();
        tb_mma<4>(acc[2 * j], acc[2 * j + 1], p, hh, hl, fr, fg);
      }
    }
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
  if (valid) {
    float* o = a.out + (size_t)row * 256;
#pragma unroll
    for (int i = 0; i < 16; ++i)
      *reinterpret_cast<float4*>(o + 16 * i + 4 * fg) = make_float4(acc[i][0], acc[i][1], acc[i][2], acc[i][3]);
  }
}

#define TB_LDS_BYTES (2 * TB_PANEL_BYTES + (TC_N + 256) * 4)

static int tb_set_attr() {
  static bool done0 = false;
  if (!done0) {
    hipError_t e = hipFuncSetAttribute((const void*)token_block_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES);
    if (e == hipSuccess)
      e = hipFuncSetAttribute((const void*)token_block_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, TB_LDS_BYTES);
    if (e != hipSuccess) {
      sam6d_set_error("token_block: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return (int)e;
    }
    done0 = true;
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
  TbArgs a{hidden, x, out, (const unsigned char*)wimage, consts, nullptr, nullptr, nullptr, M, 0, 0, 0, eps};
  hipLaunchKernelGGL(token_block_kernel<0>, dim3((unsigned)((M + TB_TOK - 1) / TB_TOK)), dim3(512), TB_LDS_BYTES, (hipStream_t)stream, a);
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
  const int tiles = (I - row0 + TB_TOK - 1) / TB_TOK;
  SAM6D_REQUIRE((long)B * tiles < 2147483647L, "linattn_layer: too many tiles");
  TbArgs a{D, nullptr, Dout, (const unsigned char*)wimage, consts, (const unsigned char*)kvimage, kvinv, ksum, 0, I, row0, tiles, eps};
  hipLaunchKernelGGL(token_block_kernel<1>, dim3((unsigned)(B * tiles)), dim3(512), TB_LDS_BYTES, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("linattn_layer");
}
