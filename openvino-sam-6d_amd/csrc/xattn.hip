// Cross attention of the sparse (197-token) transformer layers on the matrix cores, with the query projection inside
// (TransformerLayer / AttentionLayer / MultiHeadAttention of PEM/model/transformer.py:95-150):
//     q = x Wq^T + bq;   per head h:  P = softmax(q_h k_h^T / sqrt(64));   hidden[:, 64h .. 64h+64) = P v_h
// k and v come from the key-side projection: either one GEMM over the memory tokens beforehand (sam6d_cross_attention), or -- the
// default path since round 3 -- computed HERE from the memory tokens (sam6d_cross_attention_kv, template flag KVP): the workgroup of
// (cloud, head) holds Wk_h and Wv_h (64 rows x K = 256 each, 128 KiB of LDS; Wq_h and the k / v images take their place afterwards) and
// projects the 197 memory tokens of its cloud with the same register-chained transposed products; k_h / v_h never reach HBM and the
// separate (197 workgroups, launch-latency-bound) projection GEMM disappears: 12 launches per step.
//
// Before: proj_q GEMM (20 us) + a VALU attention kernel that walks the 197 keys of every query with wave butterflies (67 us), twelve
// times per step.  Here one workgroup owns a (cloud, head): the 197 x 64 keys and values of that head are cut into fp16 hi / lo halves
// ONCE into LDS images, and every 16-query group runs three register-chained, transposed products on v_mfma_f32_16x16x32_f16 (same
// scheme as block.hip: a lane holds channels of its own token, the accumulator of one product is the B operand of the next):
//     q^T (64 x tok)  = Wq_h (64 x 256)  . x^T          8 k-steps
//     S^T (keys x tok) = k_h (208 x 64)  . q^T          2 k-steps, 13 key tiles; softmax over the lane's 52 values + 3 partner lanes
//     out^T (64 x tok) = v_h^T (64 x 224) . P^T         7 k-steps
// fp16 x3 split products with power-of-two operand scales (per token row for x, q, P; per image for Wq, k, v): range-safe, ~1e-6.
#include "common.h"
#include <stdlib.h>
#include <type_traits>
#include "../../include/sam6d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

#define XA_MAXKEY 208     // 13 key tiles of 16
#define XA_NT 13
#define XA_WAVES 8
#define XA_WQ_BYTES 65536            // Wq_h image: 64 rows x 1 KiB (K = 256: hi | lo), also the v^T image (K = 224 keys + pad)
#define XA_K_BYTES (XA_MAXKEY * 256) // k_h image: 208 rows x 256 B (K = 64: hi | lo)
#define XA_LDS (XA_WQ_BYTES + XA_K_BYTES + 64)
#define XA_LDS_KV (2 * XA_WQ_BYTES + 64)  // key / value projection inside: Wk_h | Wv_h resident together, the k image overlays Wv_h

// slot p (0..31) of a 32-wide k-step <-> channel (see block.hip): channel = 16 (e >> 2) + 4 g + (e & 3), p = 8 g + e
__device__ __forceinline__ int xa_channel_slot(int c) { return 8 * ((c >> 2) & 3) + 4 * (c >> 4) + (c & 3); }

__device__ __forceinline__ float xa_pow2_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.0f;
  int e;
  (void)frexpf(amax, &e);
  e = 14 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}
__device__ __forceinline__ float xa_tok_max(float m) {
  m = fmaxf(m, xor16_f32(m));
  return fmaxf(m, xor32_f32(m));
}
__device__ __forceinline__ float xa_tok_sum(float s) {
  s += xor16_f32(s);
  return s + xor32_f32(s);
}

// acc (16 out rows x 16 tokens) += W[rows r0 .. r0+16) . X over KS k-steps.  Image rows are ROWB bytes: hi plane then lo plane (LOCH
// 16-byte chunks further), chunk c stored at (c & ~15) | ((c ^ row) & 15).
// The fragment reads run XA_FD steps ahead of the MFMAs, as inline assembly with counted waits (block.hip tb_mma does the same): left
// to the compiler every ds_read_b128 sinks to just before its use behind an lgkmcnt(0), and with three MFMAs (48 cycles) per step the
// LDS latency (~150 cycles) was exposed on every one of a wave's ~300 steps -- half of the kernel (SQ_WAIT_ANY 52 % of the wave cycles).
typedef unsigned xa_u32x4 __attribute__((ext_vector_type(4)));
#define XA_FD 4
__device__ __forceinline__ xa_u32x4 xa_lds128(unsigned addr) {
  xa_u32x4 r;
  asm volatile("ds_read_b128 %0, %1" : "=v"(r) : "v"(addr));
  return r;
}
template <int N>
__device__ __forceinline__ void xa_wait(xa_u32x4& a, xa_u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}
template <int ROWB, int LOCH, int KS, int S>
__device__ __forceinline__ void xa_mma_step(f32x4& acc, unsigned rb, int row, int fg, const half8* __restrict__ xh,
                                            const half8* __restrict__ xl, xa_u32x4 (&fh)[XA_FD + 1], xa_u32x4 (&fl)[XA_FD + 1], bool half) {
  if constexpr (S + XA_FD < KS) {
    const int ch = 4 * (S + XA_FD) + fg, cl = LOCH + 4 * (S + XA_FD) + fg;
    fh[(S + XA_FD) % (XA_FD + 1)] = xa_lds128(rb + (((ch & ~15) | ((ch ^ row) & 15)) << 4));
    fl[(S + XA_FD) % (XA_FD + 1)] = xa_lds128(rb + (((cl & ~15) | ((cl ^ row) & 15)) << 4));
  }
  constexpr int newer = (KS - 1 - S) < XA_FD ? (KS - 1 - S) : XA_FD;
  constexpr int cur = S % (XA_FD + 1);
  xa_wait<2 * newer>(fh[cur], fl[cur]);
  const half8 ah = __builtin_bit_cast(half8, fh[cur]), al = __builtin_bit_cast(half8, fl[cur]);
  if (!half) {  // (launch-uniform) matmul mode 2 keeps the hi . hi product only
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[S], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[S], acc, 0, 0, 0);
  }
  acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[S], acc, 0, 0, 0);
  if constexpr (S + 1 < KS) xa_mma_step<ROWB, LOCH, KS, S + 1>(acc, rb, row, fg, xh, xl, fh, fl, half);
}
template <int ROWB, int LOCH, int KS>
__device__ __forceinline__ void xa_mma(f32x4& acc, const unsigned char* __restrict__ img, int r0, const half8* __restrict__ xh,
                                       const half8* __restrict__ xl, int fr, int fg, bool half) {
  const int row = r0 + fr;
  const unsigned rb = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)img + (unsigned)row * ROWB;
  xa_u32x4 fh[XA_FD + 1], fl[XA_FD + 1];
#pragma unroll
  for (int s = 0; s < XA_FD && s < KS; ++s) {
    const int ch = 4 * s + fg, cl = LOCH + 4 * s + fg;
    fh[s] = xa_lds128(rb + (((ch & ~15) | ((ch ^ row) & 15)) << 4));
    fl[s] = xa_lds128(rb + (((cl & ~15) | ((cl ^ row) & 15)) << 4));
  }
  xa_mma_step<ROWB, LOCH, KS, 0>(acc, rb, row, fg, xh, xl, fh, fl, half);
}

struct XaArgs {
  const float* x;     // (B, n, 256) query-side tokens
  const float* kv;    // KVP = false: (B, m, 512) k | v of the memory tokens;  KVP = true: (B, m, 256) the memory tokens themselves
  const unsigned char* wq;  // 4 heads x 64 KiB: proj_q rows 64h .. 64h+64 as two swizzled K = 256 panels (sam6d_pack_panels)
  const float* bq;    // (256)
  float* out;         // (B, n, 256)
  int n, m;
  float inv_wq;       // 1 / pack scale of Wq
  int half;           // 1: fp16 single product (matmul mode 2)
  const unsigned char* wkv;  // KVP: 4 heads x (Wk_h 64 KiB | Wv_h 64 KiB), images like wq (sam6d_pack_panels of proj_k / proj_v rows)
  const float* bkv;          // KVP: (512) proj_k.bias | proj_v.bias
  float inv_wkv;             // KVP: 1 / pack scale of [Wk; Wv]
};

// the 16-token group `grp` of rows (b, tok, 256 channels) as split B fragments: returns the row scale
__device__ __forceinline__ float xa_load_rows(const float* __restrict__ base, int tok, half8* xh, half8* xl, int fg) {
  const float* src = base + (size_t)tok * 256;
  float4 va[8], vb4[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    va[s] = *reinterpret_cast<const float4*>(src + 32 * s + 4 * fg);
    vb4[s] = *reinterpret_cast<const float4*>(src + 32 * s + 16 + 4 * fg);
  }
  float mx = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[s].x), fabsf(va[s].y)), fmaxf(fabsf(va[s].z), fabsf(va[s].w))));
    mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb4[s].x), fabsf(vb4[s].y)), fmaxf(fabsf(vb4[s].z), fabsf(vb4[s].w))));
  }
  const float sx = xa_pow2_scale(xa_tok_max(mx));
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const float e8[8] = {va[s].x, va[s].y, va[s].z, va[s].w, vb4[s].x, vb4[s].y, vb4[s].z, vb4[s].w};
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      unsigned hi, lo;
      sam6d_split2_f16(e8[u] * sx, e8[u + 1] * sx, hi, lo);
      const _Float16 __attribute__((ext_vector_type(2))) h2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), hi);
      const _Float16 __attribute__((ext_vector_type(2))) l2 = __builtin_bit_cast(_Float16 __attribute__((ext_vector_type(2))), lo);
      xh[s][u] = h2[0];
      xh[s][u + 1] = h2[1];
      xl[s][u] = l2[0];
      xl[s][u + 1] = l2[1];
    }
  }
  return sx;
}

template <bool KVP>
__global__ __launch_bounds__(XA_WAVES * 64) void xattn_kernel(XaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* regA = lds;                    // Wq_h image, later the v_h^T image
  unsigned char* kimg = lds + XA_WQ_BYTES;      // k_h image
  float* red = reinterpret_cast<float*>(lds + (KVP ? 2 * XA_WQ_BYTES : XA_WQ_BYTES + XA_K_BYTES));  // 16 floats
  const int h = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = a.n, m = a.m;
  const bool half = a.half != 0;

  // 64 KiB image -> regA by LDS-DMA (64 pieces of 1 KiB)
  auto dma64 = [&](const unsigned char* src, unsigned char* dst) {
#pragma unroll
    for (int k = 0; k < 64 / XA_WAVES; ++k) {
      const int pc = wave + XA_WAVES * k;
      __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(src + (size_t)pc * 1024 + lane * 16),
                                       (void __attribute__((address_space(3)))*)(dst + pc * 1024), 16, 0, 0);
    }
  };
  constexpr int XA_EPT = (XA_MAXKEY * 64) / (XA_WAVES * 64);  // 26
  float kreg[KVP ? 1 : XA_EPT], vreg[KVP ? 1 : XA_EPT];
  f32x4 vacc[KVP ? 2 : 1][4];
  float sk = 0.f, sv = 0.f;
  if constexpr (KVP) {
    // ---- k_h = mem Wk_h^T + b, v_h = mem Wv_h^T + b for the m memory tokens of the cloud; a wave owns the token groups wave,
    // wave + 8 (13 groups of 16), its k / v tiles stay in registers until the workgroup-wide maxima (the images' power-of-two
    // scales) are known
    const float* mb = a.kv + (size_t)b * m * 256;
    const unsigned char* wk = a.wkv + (size_t)h * (2 * XA_WQ_BYTES);
    const int mgroups = (m + 15) >> 4;
    f32x4 kacc[2][4];
    // Wk_h -> regA, Wv_h -> the k-image region + what follows it (regB, 64 KiB): both weight images are resident while the memory
    // tokens are projected, so a token group's rows are loaded and split ONCE for its k and its v tiles
    unsigned char* regB = kimg;
    dma64(wk, regA);
    dma64(wk + XA_WQ_BYTES, regB);
    // the rows of this wave's first group are loaded and split while the weight DMA is in flight
    float mk = 0.f, mv = 0.f;
    half8 xh[8], xl[8];
    float sx = 1.0f;
    if (wave < mgroups) sx = xa_load_rows(mb, min(wave * 16 + fr, m - 1), xh, xl, fg);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int grp = wave + XA_WAVES * gi;
      if (grp < mgroups) {  // (wave-uniform)
        if (gi == 1) sx = xa_load_rows(mb, min(grp * 16 + fr, m - 1), xh, xl, fg);
        const float inv = a.inv_wkv * (1.0f / sx);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f32x4 ak = f32x4{0.f, 0.f, 0.f, 0.f}, av = f32x4{0.f, 0.f, 0.f, 0.f};
          xa_mma<1024, 32, 8>(ak, regA, 16 * i, xh, xl, fr, fg, half);
          xa_mma<1024, 32, 8>(av, regB, 16 * i, xh, xl, fr, fg, half);
          const float4 bk = *reinterpret_cast<const float4*>(a.bkv + 64 * h + 16 * i + 4 * fg);
          const float4 bv = *reinterpret_cast<const float4*>(a.bkv + 256 + 64 * h + 16 * i + 4 * fg);
          kacc[gi][i] = f32x4{ak[0] * inv + bk.x, ak[1] * inv + bk.y, ak[2] * inv + bk.z, ak[3] * inv + bk.w};
          vacc[gi][i] = f32x4{av[0] * inv + bv.x, av[1] * inv + bv.y, av[2] * inv + bv.z, av[3] * inv + bv.w};
          mk = fmaxf(mk, fmaxf(fmaxf(fabsf(kacc[gi][i][0]), fabsf(kacc[gi][i][1])), fmaxf(fabsf(kacc[gi][i][2]), fabsf(kacc[gi][i][3]))));
          mv = fmaxf(mv, fmaxf(fmaxf(fabsf(vacc[gi][i][0]), fabsf(vacc[gi][i][1])), fmaxf(fabsf(vacc[gi][i][2]), fabsf(vacc[gi][i][3]))));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) kacc[gi][i] = vacc[gi][i] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    mk = wave_max_dpp(mk);
    mv = wave_max_dpp(mv);
    if (lane == 0) { red[wave] = mk; red[8 + wave] = mv; }
    __syncthreads();  // every wave is done with both weight images
    dma64(a.wq + (size_t)h * XA_WQ_BYTES, regA);
#pragma unroll
    for (int w = 0; w < XA_WAVES; ++w) { sk = fmaxf(sk, red[w]); sv = fmaxf(sv, red[8 + w]); }
    sk = xa_pow2_scale(sk);
    sv = xa_pow2_scale(sv);
    // k_h image rows of this wave's tokens, over the Wv image (rows >= m are never read unmasked)
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int j = (wave + XA_WAVES * gi) * 16 + fr;
      if (wave + XA_WAVES * gi < mgroups) {
        // channels 16 i + 4 g + r (r = 0..3) of a key are the slots 32 (i >> 1) + 8 g + 4 (i & 1) + r of its row: four consecutive halves,
        // one 8-byte write per plane (was eight 2-byte writes)
        unsigned char* row = kimg + (size_t)j * 256;
        typedef unsigned xa_u2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned h01, l01, h23, l23;
          sam6d_split2_f16(kacc[gi][i][0] * sk, kacc[gi][i][1] * sk, h01, l01);
          sam6d_split2_f16(kacc[gi][i][2] * sk, kacc[gi][i][3] * sk, h23, l23);
          const int ch = 4 * (i >> 1) + fg, cl = 8 + ch, off = 8 * (i & 1);
          *reinterpret_cast<xa_u2*>(row + ((ch ^ (j & 15)) << 4) + off) = xa_u2{h01, h23};
          *reinterpret_cast<xa_u2*>(row + ((cl ^ (j & 15)) << 4) + off) = xa_u2{l01, l23};
        }
      }
    }
  } else {
    // ---- Wq_h image by LDS-DMA, k_h image built here from the fp32 keys
    dma64(a.wq + (size_t)h * XA_WQ_BYTES, regA);
    const float* kb = a.kv + (size_t)b * m * 512 + 64 * h;
    const float* vb = kb + 256;
    // this thread's share of the head's keys and values (element e = t + 512 i: key e >> 6, channel e & 63) goes to registers in one
    // burst of independent loads -- a loop that loads, reduces and stores element by element pays one memory round trip per element
#pragma unroll
    for (int i = 0; i < XA_EPT; ++i) {
      const int e = t + XA_WAVES * 64 * i, j = e >> 6, d = e & 63;
      const bool ok = j < m;
      kreg[i] = ok ? kb[(size_t)j * 512 + d] : 0.f;
      vreg[i] = ok ? vb[(size_t)j * 512 + d] : 0.f;
    }
    float mk = 0.f, mv = 0.f;
#pragma unroll
    for (int i = 0; i < XA_EPT; ++i) {
      mk = fmaxf(mk, fabsf(kreg[i]));
      mv = fmaxf(mv, fabsf(vreg[i]));
    }
    mk = wave_max_dpp(mk);
    mv = wave_max_dpp(mv);
    if (lane == 0) { red[wave] = mk; red[8 + wave] = mv; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < XA_WAVES; ++w) { sk = fmaxf(sk, red[w]); sv = fmaxf(sv, red[8 + w]); }
    sk = xa_pow2_scale(sk);
    sv = xa_pow2_scale(sv);
    // k_h image: every (key < 208, channel) slot is written (zeros beyond m), so no separate clearing pass
#pragma unroll
    for (int i = 0; i < XA_EPT; ++i) {
      const int e = t + XA_WAVES * 64 * i, j = e >> 6, d = e & 63;
      _Float16 hi, lo;
      sam6d_split_f16(kreg[i] * sk, hi, lo);
      const int p = 32 * (d >> 5) + xa_channel_slot(d & 31);  // half index in the hi plane (K = 64)
      _Float16* row = reinterpret_cast<_Float16*>(kimg + (size_t)j * 256);
      const int ch = p >> 3, cl = 8 + (p >> 3);
      row[((ch ^ (j & 15)) << 3) + (p & 7)] = hi;
      row[((cl ^ (j & 15)) << 3) + (p & 7)] = lo;
    }
  }

  // ---- phase 1: q^T for this wave's token groups (kept as split B fragments: 2 k-steps each)
  const int ngroups = (n + 15) >> 4;
  // gridDim.z = 2: the query groups are dealt to two workgroups per (cloud, head) -- each projects all keys / values of the head, but
  // runs one query group per wave instead of two (a cross layer of 32 clouds is 128 workgroups: half of the CUs idle otherwise)
  const int g_first = gridDim.z == 2 ? (ngroups + 1) >> 1 : ngroups;
  const int g_off = blockIdx.z ? g_first : 0, g_cnt = blockIdx.z ? ngroups - g_first : g_first;
  half8 qh[2][2], ql[2][2];
  float sq[2] = {1.f, 1.f};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();  // Wq image landed, k image complete
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int grp = g_off + wave + XA_WAVES * gi;
    if (wave + XA_WAVES * gi < g_cnt) {  // (wave-uniform)
      const int tok = min(grp * 16 + fr, n - 1);
      const float* src = a.x + ((size_t)b * n + tok) * 256;
      half8 xh[8], xl[8];
      {
        float4 va[8], vb4[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          va[s] = *reinterpret_cast<const float4*>(src + 32 * s + 4 * fg);
          vb4[s] = *reinterpret_cast<const float4*>(src + 32 * s + 16 + 4 * fg);
        }
        float mx = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(va[s].x), fabsf(va[s].y)), fmaxf(fabsf(va[s].z), fabsf(va[s].w))));
          mx = fmaxf(mx, fmaxf(fmaxf(fabsf(vb4[s].x), fabsf(vb4[s].y)), fmaxf(fabsf(vb4[s].z), fabsf(vb4[s].w))));
        }
        const float sx = xa_pow2_scale(xa_tok_max(mx));
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float e8[8] = {va[s].x, va[s].y, va[s].z, va[s].w, vb4[s].x, vb4[s].y, vb4[s].z, vb4[s].w};
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            _Float16 hi, lo;
            sam6d_split_f16(e8[u] * sx, hi, lo);
            xh[s][u] = hi;
            xl[s][u] = lo;
          }
        }
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
          xa_mma<1024, 32, 8>(acc[i], regA, 16 * i, xh, xl, fr, fg, half);
        }
        // q = acc / (s_wq sx) + bq, times 1/8 (the softmax scale 1/sqrt(64): a power of two, folded here)
        const float inv = a.inv_wq * (1.0f / sx) * 0.125f;
        float qm = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float4 bb = *reinterpret_cast<const float4*>(a.bq + 64 * h + 16 * i + 4 * fg);
          acc[i][0] = acc[i][0] * inv + bb.x * 0.125f;
          acc[i][1] = acc[i][1] * inv + bb.y * 0.125f;
          acc[i][2] = acc[i][2] * inv + bb.z * 0.125f;
          acc[i][3] = acc[i][3] * inv + bb.w * 0.125f;
          qm = fmaxf(qm, fmaxf(fmaxf(fabsf(acc[i][0]), fabsf(acc[i][1])), fmaxf(fabsf(acc[i][2]), fabsf(acc[i][3]))));
        }
        sq[gi] = xa_pow2_scale(xa_tok_max(qm));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            _Float16 hi, lo;
            sam6d_split_f16(acc[i][r] * sq[gi], hi, lo);
            qh[gi][i >> 1][4 * (i & 1) + r] = hi;
            ql[gi][i >> 1][4 * (i & 1) + r] = lo;
          }
      }
    }
  }
  __syncthreads();  // every wave is done with the Wq image

  // ---- v_h^T image over the Wq region: row d, K = key (7 k-steps of 32, padded to 8), hi plane | lo plane (32 chunks further).
  // Cleared first: the key slots 208 .. 255 are multiplied by zero probabilities only, but must not hold NaN patterns.
  for (int i = t; i < XA_WQ_BYTES / 16; i += XA_WAVES * 64) reinterpret_cast<uint4*>(regA)[i] = make_uint4(0u, 0u, 0u, 0u);
  __syncthreads();
  if constexpr (KVP) {
#pragma unroll
    for (int gi = 0; gi < 2; ++gi) {
      const int j = (wave + XA_WAVES * gi) * 16 + fr;
      if (j < m) {  // key slots >= m stay zero
        const int p = 32 * (j >> 5) + xa_channel_slot(j & 31);
        const int ch = p >> 3, cl = 32 + (p >> 3);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 16 * i + 4 * fg + r;
            _Float16 hi, lo;
            sam6d_split_f16(vacc[gi][i][r] * sv, hi, lo);
            _Float16* row = reinterpret_cast<_Float16*>(regA + (size_t)d * 1024);
            row[((((ch & ~15) | ((ch ^ d) & 15))) << 3) + (p & 7)] = hi;
            row[((((cl & ~15) | ((cl ^ d) & 15))) << 3) + (p & 7)] = lo;
          }
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < XA_EPT; ++i) {
      const int e = t + XA_WAVES * 64 * i, j = e >> 6, d = e & 63;
      _Float16 hi, lo;
      sam6d_split_f16(vreg[i] * sv, hi, lo);
      const int p = 32 * (j >> 5) + xa_channel_slot(j & 31);
      _Float16* row = reinterpret_cast<_Float16*>(regA + (size_t)d * 1024);
      const int ch = p >> 3, cl = 32 + (p >> 3);
      row[((((ch & ~15) | ((ch ^ d) & 15))) << 3) + (p & 7)] = hi;
      row[((((cl & ~15) | ((cl ^ d) & 15))) << 3) + (p & 7)] = lo;
    }
  }
  __syncthreads();

  // ---- phases 2 and 3 per token group
  const float inv_k = 1.0f / sk, inv_v = (1.0f / sv) * (1.0f / 16384.0f);
#pragma unroll
  for (int gi = 0; gi < 2; ++gi) {
    const int grp = g_off + wave + XA_WAVES * gi;
    if (wave + XA_WAVES * gi < g_cnt) {
      f32x4 s[XA_NT];
#pragma unroll
      for (int i = 0; i < XA_NT; ++i) {
        s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        xa_mma<256, 8, 2>(s[i], kimg, 16 * i, qh[gi], ql[gi], fr, fg, half);
      }
      // logits (already / 8), keys >= m masked; softmax over the token's keys
      const float inv = inv_k * (1.0f / sq[gi]);
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < XA_NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * i + 4 * fg + r;
          s[i][r] = key < m ? s[i][r] * inv : -INFINITY;
          mx = fmaxf(mx, s[i][r]);
        }
      mx = xa_tok_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < XA_NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[i][r] = __builtin_amdgcn_exp2f((s[i][r] - mx) * 1.4426950408889634f);
          sum += s[i][r];
        }
      sum = xa_tok_sum(sum);
      const float pscale = 16384.0f / sum;  // probabilities times 2^14 (<= 2^14: fp16-safe), the 2^-14 is in inv_v
      half8 ph[7], pl[7];
#pragma unroll
      for (int i = 0; i < 14; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = i < XA_NT ? s[i < XA_NT ? i : 0][r] * pscale : 0.f;
          _Float16 hi, lo;
          sam6d_split_f16(pv, hi, lo);
          ph[i >> 1][4 * (i & 1) + r] = hi;
          pl[i >> 1][4 * (i & 1) + r] = lo;
        }
      f32x4 o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        xa_mma<1024, 32, 7>(o[i], regA, 16 * i, ph, pl, fr, fg, half);
      }
      const int tok = grp * 16 + fr;
      if (tok < n) {
        float* dst = a.out + ((size_t)b * n + tok) * 256 + 64 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(dst + 16 * i + 4 * fg) = make_float4(o[i][0] * inv_v, o[i][1] * inv_v, o[i][2] * inv_v, o[i][3] * inv_v);
      }
    }
  }
}

static int xa_cu_count[SAM6D_MAX_DEVICES];
// query split of a launch: two workgroups per (cloud, head) while they all fit the chip at once (the results do not depend on it)
static int xa_qsplit(int B, int n) {
  static int allow = -1;  // SAM6D_XATTN_QSPLIT=0: always one workgroup per (cloud, head) (A/B runs)
  if (allow < 0) {
    const char* e = getenv("SAM6D_XATTN_QSPLIT");
    allow = (e && e[0] == '0') ? 0 : 1;
  }
  if (!allow) return 1;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SAM6D_MAX_DEVICES) return 1;
  const int cu = xa_cu_count[dev];
  return (cu > 0 && 8 * B <= cu && n > 16) ? 2 : 1;
}
static int xa_reserve() {
  static unsigned long long done = 0;
  int dev_ = 0;
  if (sam6d_first_use_on_device(&done, &dev_)) {
    if (dev_ >= 0 && dev_ < SAM6D_MAX_DEVICES) {
      int cu = 0;
      if (hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev_) == hipSuccess) xa_cu_count[dev_] = cu;
    }
    hipError_t e = hipFuncSetAttribute((const void*)xattn_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)xattn_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS_KV);
    if (e != hipSuccess) {
      sam6d_set_error("cross_attention: cannot reserve %d bytes of LDS: %s", XA_LDS, hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&done);
  }
  return 0;
}

extern "C" int sam6d_cross_attention(const float* x, const float* kv, const void* wq_image, const float* bq, float inv_wq_scale,
                                     float* out, int B, int n, int m, void* stream) {
  SAM6D_REQUIRE(x && kv && wq_image && bq && out && B >= 0, "cross_attention: null pointer");
  SAM6D_REQUIRE(n > 0 && n <= 16 * 2 * XA_WAVES && m > 0 && m <= XA_MAXKEY, "cross_attention: needs n <= %d queries and m <= %d keys per cloud",
                16 * 2 * XA_WAVES, XA_MAXKEY);
  SAM6D_REQUIRE(((((size_t)x) | ((size_t)kv) | ((size_t)wq_image) | ((size_t)bq) | ((size_t)out)) & 15) == 0,
                "cross_attention: pointers must be 16-byte aligned");
  SAM6D_REQUIRE(B <= 65535 && inv_wq_scale > 0.f, "cross_attention: bad arguments");
  if (B == 0) return 0;
  if (int rc = xa_reserve()) return rc;
  XaArgs a{x, kv, (const unsigned char*)wq_image, bq, out, n, m, inv_wq_scale, sam6d_half_for(2), nullptr, nullptr, 1.0f};
  hipLaunchKernelGGL(xattn_kernel<false>, dim3(4, B, xa_qsplit(B, n)), dim3(XA_WAVES * 64), XA_LDS, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("cross_attention");
}

extern "C" long sam6d_cross_attention_kv_image_bytes(void) { return 4L * 2 * XA_WQ_BYTES; }

extern "C" int sam6d_cross_attention_kv(const float* x, const float* mem, const void* wq_image, const float* bq, float inv_wq_scale,
                                        const void* wkv_image, const float* bkv, float inv_wkv_scale, float* out, int B, int n, int m,
                                        void* stream) {
  SAM6D_REQUIRE(x && mem && wq_image && bq && wkv_image && bkv && out && B >= 0, "cross_attention_kv: null pointer");
  SAM6D_REQUIRE(n > 0 && n <= 16 * 2 * XA_WAVES && m > 0 && m <= XA_MAXKEY,
                "cross_attention_kv: needs n <= %d queries and m <= %d keys per cloud", 16 * 2 * XA_WAVES, XA_MAXKEY);
  SAM6D_REQUIRE(((((size_t)x) | ((size_t)mem) | ((size_t)wq_image) | ((size_t)bq) | ((size_t)wkv_image) | ((size_t)bkv) | ((size_t)out)) & 15) == 0,
                "cross_attention_kv: pointers must be 16-byte aligned");
  SAM6D_REQUIRE(B <= 65535 && inv_wq_scale > 0.f && inv_wkv_scale > 0.f, "cross_attention_kv: bad arguments");
  if (B == 0) return 0;
  if (int rc = xa_reserve()) return rc;
  XaArgs a{x, mem, (const unsigned char*)wq_image, bq, out, n, m, inv_wq_scale, sam6d_half_for(2), (const unsigned char*)wkv_image, bkv,
           inv_wkv_scale};
  hipLaunchKernelGGL(xattn_kernel<true>, dim3(4, B, xa_qsplit(B, n)), dim3(XA_WAVES * 64), XA_LDS_KV, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("cross_attention_kv");
}

// ---------------------------------------------------------------------------------------------------------------------
// Tail of the RPE self-attention (RPEMultiHeadAttention.forward, PEM/model/transformer.py:405-416) once q, k, v exist (rpe_front)
// and the geometric score term G[q][h][m] = q_h . proj_p(E[q, m])_h has been rebuilt (rpe_score_kernel, raw mode; listed pairs added
// by rpe_listed_kernel):   P = softmax((q_h k_h^T + G) / 8) over the keys;   hidden[:, 64h .. 64h+64) = P v_h.
// One workgroup per (cloud, head), the cross-attention kernel above without its projections: k_h and v_h^T become fp16 hi / lo LDS
// images once, every 16-query group runs S^T = k_h q^T (2 k-steps x 13 key tiles), adds its G tile (one float4 per lane and tile:
// the accumulator holds keys 16 i + 4 g + r of the lane's own query), softmax, out^T = v_h^T P^T (7 k-steps).
// Before: a 4096-workgroup batched GEMM for q.k^T (35 us), the softmax inside the score kernel, and a second batched GEMM for P.v
// (22 us) -- both launch-latency-bound at 197 x 197 x 64 per (cloud, head).
#ifdef SA_STAMP  // diagnostic build only (scratch/sa_stamps.py): per-wave s_memtime stamps of sattn_kernel; no output depends on them
__device__ unsigned long long sa_stamps[256 * 8 * 16];
extern "C" int sam6d_sattn_debug_stamps(void* dst) {
  return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(sa_stamps), sizeof(sa_stamps));
}
#define SA_ST(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define SA_ST(i)
#endif
struct SaArgs {
  const float* qkv;   // (B n, 768): q | k | v
  const float* G;     // (B n, 4, ldp) geometric score term (raw, not yet / 8)
  float* out;         // (B n, 256)
  int n, ldp;
  int half;
};

// S^T tiles 0 .. NT-1 (16 keys each, K = 64 = 2 k-steps) as ONE stream of 2 NT steps through a fragment ring: the reads of step
// S + XA_FD go out before the MFMAs of step S.  (xa_mma per tile issues its reads only after the previous tile's MFMAs: stamps put
// the 13 tiles of a group at ~7 000 cycles for 1 250 cycles of MFMA issue.)
#define XA_FDR 4  // ring depth of the score-tile stream
template <int NT, int S>
__device__ __forceinline__ void xa_rows_step(f32x4* acc, unsigned kb, int fr, int fg, const half8* __restrict__ xh,
                                             const half8* __restrict__ xl, xa_u32x4 (&fh)[XA_FDR + 1], xa_u32x4 (&fl)[XA_FDR + 1], bool half) {
  auto issue = [&](int st) {
    const int row = 16 * (st >> 1) + fr, ks = st & 1;
    const int ch = 4 * ks + fg, cl = 8 + 4 * ks + fg;
    const unsigned rb = kb + (unsigned)row * 256;
    fh[st % (XA_FDR + 1)] = xa_lds128(rb + ((((ch & ~15) | ((ch ^ row) & 15))) << 4));
    fl[st % (XA_FDR + 1)] = xa_lds128(rb + ((((cl & ~15) | ((cl ^ row) & 15))) << 4));
  };
  if constexpr (S == 0) {
#pragma unroll
    for (int st = 0; st < XA_FDR && st < 2 * NT; ++st) issue(st);
  }
  if constexpr (S + XA_FDR < 2 * NT) issue(S + XA_FDR);
  constexpr int newer = (2 * NT - 1 - S) < XA_FDR ? (2 * NT - 1 - S) : XA_FDR;
  constexpr int cur = S % (XA_FDR + 1);
  xa_wait<2 * newer>(fh[cur], fl[cur]);
  const half8 ah = __builtin_bit_cast(half8, fh[cur]), al = __builtin_bit_cast(half8, fl[cur]);
  f32x4& c = acc[S >> 1];
  if (!half) {
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[S & 1], c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[S & 1], c, 0, 0, 0);
  }
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[S & 1], c, 0, 0, 0);
  if constexpr (S + 1 < 2 * NT) xa_rows_step<NT, S + 1>(acc, kb, fr, fg, xh, xl, fh, fl, half);
}

__global__ __launch_bounds__(XA_WAVES * 64) void sattn_kernel(SaArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* regA = lds;                    // v_h^T image
  unsigned char* kimg = lds + XA_WQ_BYTES;      // k_h image
  float* red = reinterpret_cast<float*>(lds + XA_WQ_BYTES + XA_K_BYTES);  // 16 floats
  const int h = blockIdx.x, b = blockIdx.y;
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), fr = lane & 15, fg = lane >> 4;
  const int n = a.n, m = a.n;
  const bool half = a.half != 0;
  const int ngroups = (n + 15) >> 4;
#ifdef SA_STAMP
  unsigned long long st[16];
  for (int i = 0; i < 16; ++i) st[i] = 0;
#endif
  SA_ST(0);
  // ---- everything this wave will read from global memory is requested up front (stamps: the score-term tiles alone, requested where
  // they are used, cost each group ~10 000 cycles of exposed latency, and 52 scalar element loads per thread 10 000 cycles of issue):
  // the score-term tiles and query rows of the first token group (the second group's during the first one's softmax), the head's keys as float4 (key j, channels 4 c .. 4 c + 3), its values
  // as 4 scalars (channel d, keys 4 q .. 4 q + 3: the transposed image wants consecutive keys of one channel)
  float4 gt[XA_NT];  // the score-term tiles of the group at hand (group 1's are requested while group 0's softmax runs)
  float4 qa[2][2], qb[2][2];
  // G tiles of the lane's query: keys 16 i + 4 g .. + 3 (a row of ldp = 4 ceil(n / 4) floats, so a started float4 stays inside it)
  auto load_g = [&](int gi) {
    const int grp = wave + XA_WAVES * gi;
    const int tok = min(grp * 16 + fr, n - 1);
    const float* grow = a.G + (((size_t)b * n + tok) * 4 + h) * a.ldp + 4 * fg;
#pragma unroll
    for (int i = 0; i < XA_NT; ++i)
      gt[i] = (grp < ngroups && 16 * i + 4 * fg < m) ? *reinterpret_cast<const float4*>(grow + 16 * i) : make_float4(0.f, 0.f, 0.f, 0.f);
  };
  auto load_q = [&](int gi) {
    const int grp = wave + XA_WAVES * gi;
    const int tok = min(grp * 16 + fr, n - 1);
    const float* src = a.qkv + ((size_t)b * n + tok) * 768 + 64 * h;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      qa[gi][s2] = *reinterpret_cast<const float4*>(src + 32 * s2 + 4 * fg);
      qb[gi][s2] = *reinterpret_cast<const float4*>(src + 32 * s2 + 16 + 4 * fg);
    }
  };
  constexpr int XA_KT = (XA_MAXKEY * 16 + XA_WAVES * 64 - 1) / (XA_WAVES * 64);        // 7 float4 of k per thread
  constexpr int XA_VT = ((XA_MAXKEY / 4) * 64 + XA_WAVES * 64 - 1) / (XA_WAVES * 64);  // 7 key quads of v per thread
  const float* kb = a.qkv + (size_t)b * n * 768 + 256 + 64 * h;
  const float* vb = kb + 256;
  float4 kreg[XA_KT];
  float vreg[XA_VT][4];
#pragma unroll
  for (int i = 0; i < XA_KT; ++i) {
    const int e = t + XA_WAVES * 64 * i, j = e >> 4, c4 = e & 15;
    kreg[i] = (j < m) ? *reinterpret_cast<const float4*>(kb + (size_t)j * 768 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int i = 0; i < XA_VT; ++i) {
    const int e = t + XA_WAVES * 64 * i, d = e & 63, jq = e >> 6;  // a wave reads 64 consecutive channels of one key
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * jq + r;
      vreg[i][r] = (j < m) ? vb[(size_t)j * 768 + d] : 0.f;
    }
  }
  // (behind the key / value loads: the vector-memory counter retires in order, and the images are built first)
  load_g(0);
  load_q(0);
  // the v^T image is cleared first: its key slots >= 208 are multiplied by zero probabilities only, but must not hold NaN patterns
  for (int i = t; i < XA_WQ_BYTES / 16; i += XA_WAVES * 64) reinterpret_cast<uint4*>(regA)[i] = make_uint4(0u, 0u, 0u, 0u);
  float mk = 0.f, mv = 0.f;
#pragma unroll
  for (int i = 0; i < XA_KT; ++i) mk = fmaxf(mk, fmaxf(fmaxf(fabsf(kreg[i].x), fabsf(kreg[i].y)), fmaxf(fabsf(kreg[i].z), fabsf(kreg[i].w))));
#pragma unroll
  for (int i = 0; i < XA_VT; ++i) mv = fmaxf(mv, fmaxf(fmaxf(fabsf(vreg[i][0]), fabsf(vreg[i][1])), fmaxf(fabsf(vreg[i][2]), fabsf(vreg[i][3]))));
  mk = wave_max_dpp(mk);
  mv = wave_max_dpp(mv);
  SA_ST(1);
  if (lane == 0) { red[wave] = mk; red[8 + wave] = mv; }
  __syncthreads();
  SA_ST(2);
  float sk = 0.f, sv = 0.f;
#pragma unroll
  for (int w = 0; w < XA_WAVES; ++w) { sk = fmaxf(sk, red[w]); sv = fmaxf(sv, red[8 + w]); }
  sk = xa_pow2_scale(sk);
  sv = xa_pow2_scale(sv);
  typedef unsigned sa_u2 __attribute__((ext_vector_type(2)));
  // k_h image: channels 4 c .. 4 c + 3 of key j are four consecutive slots (8 bytes) of its row; every (key < 208, channel) slot is
  // written (zeros beyond m)
#pragma unroll
  for (int i = 0; i < XA_KT; ++i) {
    const int e = t + XA_WAVES * 64 * i, j = e >> 4, c4 = e & 15;
    if (j < XA_MAXKEY) {
      unsigned h01, l01, h23, l23;
      sam6d_split2_f16(kreg[i].x * sk, kreg[i].y * sk, h01, l01);
      sam6d_split2_f16(kreg[i].z * sk, kreg[i].w * sk, h23, l23);
      const int ch = 4 * (c4 >> 3) + (c4 & 3), cl = 8 + ch, off = ((c4 >> 2) & 1) * 8;  // slot 32 (d >> 5) + 8 ((d >> 2) & 3) + 4 ((d >> 4) & 1) + (d & 3)
      unsigned char* row = kimg + (size_t)j * 256;
      *reinterpret_cast<sa_u2*>(row + ((ch ^ (j & 15)) << 4) + off) = sa_u2{h01, h23};
      *reinterpret_cast<sa_u2*>(row + ((cl ^ (j & 15)) << 4) + off) = sa_u2{l01, l23};
    }
  }
  // v_h^T image: row d, K = key; keys 4 q .. 4 q + 3 are four consecutive slots
#pragma unroll
  for (int i = 0; i < XA_VT; ++i) {
    const int e = t + XA_WAVES * 64 * i, d = e & 63, jq = e >> 6;
    if (4 * jq < XA_MAXKEY) {
      unsigned h01, l01, h23, l23;
      sam6d_split2_f16(vreg[i][0] * sv, vreg[i][1] * sv, h01, l01);
      sam6d_split2_f16(vreg[i][2] * sv, vreg[i][3] * sv, h23, l23);
      const int j0 = 4 * jq, p = 32 * (j0 >> 5) + xa_channel_slot(j0 & 31);
      const int ch = p >> 3, cl = 32 + (p >> 3), off = (p & 7) * 2;
      unsigned char* row = regA + (size_t)d * 1024;
      *reinterpret_cast<sa_u2*>(row + ((((ch & ~15) | ((ch ^ d) & 15))) << 4) + off) = sa_u2{h01, h23};
      *reinterpret_cast<sa_u2*>(row + ((((cl & ~15) | ((cl ^ d) & 15))) << 4) + off) = sa_u2{l01, l23};
    }
  }
  SA_ST(3);
  __syncthreads();
  SA_ST(4);

  const float inv_k = 1.0f / sk, inv_v = (1.0f / sv) * (1.0f / 16384.0f);
  const unsigned kimg_lds = (unsigned)(size_t)(__attribute__((address_space(3))) const unsigned char*)kimg;
  auto group = [&](auto GI) {
    constexpr int gi = decltype(GI)::value;
    const int grp = wave + XA_WAVES * gi;
    if (grp < ngroups) {  // (wave-uniform)
      const int tok = min(grp * 16 + fr, n - 1);
      // q / 8 (the softmax scale 1 / sqrt(64): a power of two) as split B fragments
      half8 qh[2], ql[2];
      float qm = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float4 u = qa[gi][s2], w = qb[gi][s2];
        qm = fmaxf(qm, fmaxf(fmaxf(fabsf(u.x), fabsf(u.y)), fmaxf(fabsf(u.z), fabsf(u.w))));
        qm = fmaxf(qm, fmaxf(fmaxf(fabsf(w.x), fabsf(w.y)), fmaxf(fabsf(w.z), fabsf(w.w))));
      }
      const float sq = xa_pow2_scale(xa_tok_max(qm * 0.125f));
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const float e8[8] = {qa[gi][s2].x, qa[gi][s2].y, qa[gi][s2].z, qa[gi][s2].w, qb[gi][s2].x, qb[gi][s2].y, qb[gi][s2].z, qb[gi][s2].w};
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          _Float16 hi, lo;
          sam6d_split_f16(e8[u] * 0.125f * sq, hi, lo);
          qh[s2][u] = hi;
          ql[s2][u] = lo;
        }
      }
      f32x4 s[XA_NT];
#pragma unroll
      for (int i = 0; i < XA_NT; ++i) s[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      SA_ST(5 + 5 * gi);
      {
        xa_u32x4 fh[XA_FDR + 1], fl[XA_FDR + 1];
        xa_rows_step<XA_NT, 0>(s, kimg_lds, fr, fg, qh, ql, fh, fl, half);
      }
      SA_ST(6 + 5 * gi);
      const float inv = inv_k * (1.0f / sq);
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < XA_NT; ++i) {
        const float gv[4] = {gt[i].x, gt[i].y, gt[i].z, gt[i].w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * i + 4 * fg + r;
          s[i][r] = key < m ? s[i][r] * inv + gv[r] * 0.125f : -INFINITY;
          mx = fmaxf(mx, s[i][r]);
        }
      }
      mx = xa_tok_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int i = 0; i < XA_NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          s[i][r] = __builtin_amdgcn_exp2f((s[i][r] - mx) * 1.4426950408889634f);
          sum += s[i][r];
        }
      sum = xa_tok_sum(sum);
      const float pscale = 16384.0f / sum;  // probabilities times 2^14 (fp16-safe), the 2^-14 is in inv_v
      half8 ph[7], pl[7];
#pragma unroll
      for (int i = 0; i < 14; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = i < XA_NT ? s[i < XA_NT ? i : 0][r] * pscale : 0.f;
          _Float16 hi, lo;
          sam6d_split_f16(pv, hi, lo);
          ph[i >> 1][4 * (i & 1) + r] = hi;
          pl[i >> 1][4 * (i & 1) + r] = lo;
        }
      if (gi == 0) {  // the next group's inputs: in flight under this group's P.v and store and the next group's score tiles
        load_g(1);      // (requested earlier, beside the logits and the probability halves, they do not fit the register file)
        load_q(1);
      }
      f32x4 o[4];
      SA_ST(7 + 5 * gi);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        xa_mma<1024, 32, 7>(o[i], regA, 16 * i, ph, pl, fr, fg, half);
      }
      SA_ST(8 + 5 * gi);
      if (grp * 16 + fr < n) {
        float* dst = a.out + ((size_t)b * n + tok) * 256 + 64 * h;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          *reinterpret_cast<float4*>(dst + 16 * i + 4 * fg) = make_float4(o[i][0] * inv_v, o[i][1] * inv_v, o[i][2] * inv_v, o[i][3] * inv_v);
      }
      SA_ST(9 + 5 * gi);
    }
  };
  group(std::integral_constant<int, 0>{});
  group(std::integral_constant<int, 1>{});
#ifdef SA_STAMP
  st[15] = __builtin_amdgcn_s_memtime();
  if (lane == 0 && blockIdx.y * 4 + blockIdx.x < 256)
    for (int i = 0; i < 16; ++i) sa_stamps[((blockIdx.y * 4 + blockIdx.x) * 8 + wave) * 16 + i] = st[i];
#endif
}

extern "C" int sam6d_rpe_self_attention(const float* qkv, const float* G, float* hidden, int B, int n, int ldp, void* stream) {
  SAM6D_REQUIRE(qkv && G && hidden && B >= 0, "rpe_self_attention: null pointer");
  SAM6D_REQUIRE(n > 0 && n <= XA_MAXKEY && ldp >= n && (ldp & 3) == 0,
                "rpe_self_attention: needs n <= %d tokens per cloud and ldp >= n a multiple of 4 (n = %d, ldp = %d)", XA_MAXKEY, n, ldp);
  SAM6D_REQUIRE(((((size_t)qkv) | ((size_t)G) | ((size_t)hidden)) & 15) == 0, "rpe_self_attention: pointers must be 16-byte aligned");
  SAM6D_REQUIRE(B <= 65535, "rpe_self_attention: B <= 65535");
  if (B == 0) return 0;
  static unsigned long long done = 0;
  if (sam6d_first_use_on_device(&done)) {
    hipError_t e = hipFuncSetAttribute((const void*)sattn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, XA_LDS);
    if (e != hipSuccess) {
      sam6d_set_error("rpe_self_attention: cannot reserve %d bytes of LDS: %s", XA_LDS, hipGetErrorString(e));
      return (int)e;
    }
    sam6d_setup_done_on_device(&done);
  }
  SaArgs a{qkv, G, hidden, n, ldp, sam6d_half_for(2)};
  hipLaunchKernelGGL(sattn_kernel, dim3(4, B), dim3(XA_WAVES * 64), XA_LDS, (hipStream_t)stream, a);
  SAM6D_LAUNCH_CHECK("rpe_self_attention");
}
