// Shared helpers for the gfx950 (CDNA4, wave64) kernels of libsam6d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define SAM6D_WAVE 64

// ---- error plumbing -------------------------------------------------------------------------------------
// Every extern "C" entry point returns 0 on success, a negative SAM6D_E* code for bad arguments and a positive
// hipError_t for runtime failures; sam6d_last_error() returns the text.  (The reference prints and exit(-1)s on a
// CUDA launch failure, EXT/include/cuda_utils.h:42-51; SURVEY 8b asks for a checked error instead.)
#define SAM6D_EINVAL (-1)
#define SAM6D_ENOTIMPL (-2)

void sam6d_set_error(const char* fmt, ...);

#define SAM6D_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      sam6d_set_error(__VA_ARGS__);         \
      return SAM6D_EINVAL;                  \
    }                                       \
  } while (0)

#define SAM6D_LAUNCH_CHECK(name)                                                        \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      sam6d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return (int)e__;                                                                  \
    }                                                                                   \
    return 0;                                                                           \
  } while (0)

#define SAM6D_LAUNCH_CHECK_CONT(name)                                                   \
  do {                                                                                  \
    hipError_t e__ = hipGetLastError();                                                 \
    if (e__ != hipSuccess) {                                                            \
      sam6d_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));           \
      return (int)e__;                                                                  \
    }                                                                                   \
  } while (0)

// ---- device helpers ---------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// ---- DPP / permlane reductions (no LDS crossbar: ds_bpermute costs an LDS round trip per step, these are plain VALU).
// The summation ORDER differs from the xor butterfly above, so results can differ in the last bit: use them where no bit
// recipe of the reference is pinned (attention, LayerNorm, focus, norms); max / min are order-independent.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float x) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x), CTRL, 0xf, 0xf, false));
}
// all-reduce over each 16-lane row: rotations by 8, 4, 2, 1 (row_ror) -- every lane ends with the row total
__device__ __forceinline__ float row16_sum_dpp(float v) {
  v += dpp_f32<0x128>(v);  // row_ror:8
  v += dpp_f32<0x124>(v);  // row_ror:4
  v += dpp_f32<0x122>(v);  // row_ror:2
  v += dpp_f32<0x121>(v);  // row_ror:1
  return v;
}
__device__ __forceinline__ float row16_max_dpp(float v) {
  v = fmaxf(v, dpp_f32<0x128>(v));
  v = fmaxf(v, dpp_f32<0x124>(v));
  v = fmaxf(v, dpp_f32<0x122>(v));
  v = fmaxf(v, dpp_f32<0x121>(v));
  return v;
}
// exchange with the lane 16 / 32 away (gfx950 v_permlane16_swap / v_permlane32_swap on a copy)
__device__ __forceinline__ float xor16_f32(float v) {
  typedef unsigned u2_ __attribute__((ext_vector_type(2)));
  const u2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 16) ? r[0] : r[1]);
}
__device__ __forceinline__ float xor32_f32(float v) {
  typedef unsigned u2_ __attribute__((ext_vector_type(2)));
  const u2_ r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
  v = row16_sum_dpp(v);
  v += xor16_f32(v);
  v += xor32_f32(v);
  return v;
}
__device__ __forceinline__ float wave_max_dpp(float v) {
  v = row16_max_dpp(v);
  v = fmaxf(v, xor16_f32(v));
  v = fmaxf(v, xor32_f32(v));
  return v;
}

__device__ __forceinline__ float row16_min_dpp(float v) {
  v = fminf(v, dpp_f32<0x128>(v));
  v = fminf(v, dpp_f32<0x124>(v));
  v = fminf(v, dpp_f32<0x122>(v));
  v = fminf(v, dpp_f32<0x121>(v));
  return v;
}
__device__ __forceinline__ float wave_min_dpp(float v) {
  v = row16_min_dpp(v);
  v = fminf(v, xor16_f32(v));
  v = fminf(v, xor32_f32(v));
  return v;
}
// max / min do not depend on the order of the reduction: the DPP form everywhere (round 4; the ds_bpermute butterfly cost six LDS round
// trips per call in the latency-bound per-proposal kernels)
__device__ __forceinline__ float wave_max(float v) { return wave_max_dpp(v); }
__device__ __forceinline__ float wave_min(float v) { return wave_min_dpp(v); }
// first maximum of (value, index) pairs over the wave: the largest value and, among the lanes that hold it, the lowest index (what a
// sequential `v > best` scan in index order keeps).  Indices below 2^24 (exact as floats); index 0x7fffffff = "no element seen".
__device__ __forceinline__ void wave_argmax_first(float& best, int& bi) {
  const float m = wave_max_dpp(best);
  const float nk = wave_max_dpp((best == m && bi != 0x7fffffff) ? -(float)bi : -3.0e38f);
  best = m;
  bi = nk < -1.0e30f ? 0x7fffffff : (int)(-nk);
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long w = __shfl_xor(v, o, 64);
    v = w > v ? w : v;
  }
  return v;
}

// squared distance in the torch-CPU K=3 matmul recipe (PEM/utils/model_utils.py:101-128):
//   s = (p0*p0 + p1*p1) + p2*p2 (plain adds), xy = fma(x2,y2, fma(x1,y1, x0*y0)), d = max((sx - 2xy) + sy, 0)
// The library is built with -ffp-contract=off, so only the explicit fmaf calls fuse.
__device__ __forceinline__ float sqnorm3(float a, float b, float c) { return (a * a + b * b) + c * c; }
__device__ __forceinline__ float pdist3(float x0, float x1, float x2, float sx, float y0, float y1, float y2, float sy) {
  const float xy = fmaf(x2, y2, fmaf(x1, y1, x0 * y0));
  const float d = (sx - 2.0f * xy) + sy;
  return d < 0.0f ? 0.0f : d;
}

// sin and cos of the same fp32 argument, <= 1.5 ulp each for |x| < 1e5 (checked against fp64 over [0, 1000], the range of
// the sinusoidal-embedding arguments: indices up to ~870 times frequencies <= 1): three-term Cody-Waite reduction by
// pi/2 with fma, degree-9 / degree-8 minimax polynomials on [-pi/4, pi/4], quadrant fix-up.  ~30 VALU instructions,
// branch-free.  The caller guarantees the range (geo.hip routes clouds with larger indices to the sincosf kernel).
#define SAM6D_FAST_SINCOS_LIMIT 1.0e5f
__device__ __forceinline__ void fast_sincosf(float x, float* sn, float* cs) {
  const float j = rintf(x * 0.636619747f);
  float a = fmaf(j, -1.57079601e+00f, x);
  a = fmaf(j, -3.13916473e-07f, a);
  a = fmaf(j, -5.39030253e-15f, a);
  const float s = a * a;
  float z = 2.86567956e-6f;
  z = fmaf(z, s, -1.98559923e-4f);
  z = fmaf(z, s, 8.33338592e-3f);
  z = fmaf(z, s, -1.66666672e-1f);
  z = z * s;
  const float ps = fmaf(z, a, a);
  float c = 2.44677067e-5f;
  c = fmaf(c, s, -1.38877297e-3f);
  c = fmaf(c, s, 4.16666567e-2f);
  c = fmaf(c, s, -0.5f);
  const float pc = fmaf(c, s, 1.0f);
  const int q = (int)j;
  const float s0 = (q & 1) ? pc : ps, c0 = (q & 1) ? ps : pc;
  *sn = (q & 2) ? -s0 : s0;
  *cs = ((q + 1) & 2) ? -c0 : c0;
}

// fp32 -> fp16 (hi, lo) pair with hi + lo = x to 22 significand bits.  x is first made opaque to the optimiser: when x is the product
// of a multiplication, clang folds the multiplication into ONE of the conversions (v_fma_mixlo_f16: a single rounding of the exact
// product) and uses the separately rounded fp32 product for the other; on a rounding tie the two disagree and hi + lo is off by one
// fp16 ulp of hi (found as a 3e-5 error of a single softmax probability, 2^-11 relative).  The empty asm costs no instruction.
__device__ __forceinline__ void sam6d_split_f16(float x, _Float16& hi, _Float16& lo) {
  asm("" : "+v"(x));
  hi = (_Float16)x;
  lo = (_Float16)(x - (float)hi);
}

// The same split for a pair, as four instructions: v_cvt_pk_f16_f32 (hi pair, round to nearest even), two v_fma_mix_f32 (x - hi, exact: the
// fp16 operand is read straight from the packed register) and v_cvt_pk_f16_f32 (lo pair).  The plain C form above costs eight (two
// conversions, two conversions back, two subtractions, a conversion pair, a pack); every split-precision kernel splits activations between
// its MFMAs, where vector instructions are what bounds it (profiles/README.md, issue micro-benchmark).  Bit-identical results.
__device__ __forceinline__ void sam6d_split2_f16(float a, float b, unsigned& hi2, unsigned& lo2) {
  float la, lb;
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi2) : "v"(a), "v"(b));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi2), "v"(a));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi2), "v"(b));
  asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lo2) : "v"(la), "v"(lb));
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// One-time per-DEVICE setup (hipFuncSetAttribute, CU count ...): `done` is a bit mask over device ordinals owned by the call site.
// sam6d_first_use_on_device returns true while the calling thread's current device has not COMPLETED its setup; the call site runs the
// setup and, only after it succeeded, calls sam6d_setup_done_on_device -- a failed setup is retried by the next call instead of being
// skipped.  The mask is read / updated atomically: two host threads driving different GPUs may both run the (idempotent) setup of
// their own device, neither can lose the other's bit.  A process that drives several GPUs gets every device initialised, instead of
// device 0's state being reused for all of them.  Ordinals >= SAM6D_MAX_DEVICES: dev_out = -1, the call site rejects the launch.
#define SAM6D_MAX_DEVICES 64
static inline bool sam6d_first_use_on_device(const unsigned long long* done, int* dev_out = nullptr) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (dev < 0 || dev >= SAM6D_MAX_DEVICES) {
    if (dev_out) *dev_out = -1;
    return true;  // never cached: the setup runs (or the call site rejects the ordinal) on every call
  }
  if (dev_out) *dev_out = dev;
  return ((__atomic_load_n(done, __ATOMIC_ACQUIRE) >> dev) & 1ull) == 0;
}
static inline void sam6d_setup_done_on_device(unsigned long long* done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SAM6D_MAX_DEVICES) return;
  (void)__atomic_fetch_or(done, 1ull << dev, __ATOMIC_RELEASE);
}

// matmul mode 2 (fp16 single product) per kernel family: bit 0 generic GEMM, 1 block kernels, 2 cross attention, 3 fine similarity.
// SAM6D_HALF_MASK (environment, read once; default 15 = all) narrows it -- a bisecting aid, see DESIGN "Mode 2".
extern "C" int sam6d_get_matmul_mode(void);
static inline int sam6d_half_for(int family_bit) {
  static int mask = -1;
  if (mask < 0) {
    const char* e = getenv("SAM6D_HALF_MASK");
    mask = e ? atoi(e) : 15;
  }
  return (sam6d_get_matmul_mode() == 2 && ((mask >> family_bit) & 1)) ? 1 : 0;
}
