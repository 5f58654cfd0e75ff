// Point-cloud primitives of the PEM matching path for gfx950: farthest-point sampling, index gathers, ball query,
// grouping.  These are the device side of the reference's `pointnet2._ext` seam (EXT/src/bindings.cpp:11-24) and
// reproduce the reference's CPU loops bit-for-bit (EXT/src/sampling.cpp:76-118, ball_query.cpp:16-62,
// group_points.cpp:20-45); the CUDA kernels in EXT/src/*_gpu.cu differ from those loops (no origin skip, other tie
// order) and are NOT the contract.
//
// HBM-bound / latency-bound integer+fp32 work: no MFMA here.  Layout notes per kernel.
#include "common.h"
#include "../../include/sam6d_hip.h"

#include <float.h>
#include <stdint.h>

// =========================================================================================================
// Farthest point sampling.
// One workgroup per cloud (the selection is sequential in j).  Points live in LDS (for the broadcast of the last
// selected point) and each thread keeps its PPT points + running min-distance in registers.  Per round:
// PPT distance updates -> thread-local arg-max -> wave reduction -> one LDS slot per wave -> one barrier -> every thread reduces the
// NW wave slots.
//
// Selection rule: max distance, lowest index on ties == the reference's strict `d2 > best` scan in increasing k (sampling.cpp:105-111).
// (fps_big_kernel / fps_grid_kernel pack it into one 64-bit key: d2 >= +0 so its bit pattern is monotone; key = bits(d2) << 32 | ~idx.)
// Points inside the origin ball (mag <= 1e-3, compared in double like the reference, sampling.cpp:102-103) never
// compete and keep temp = FLT_MAX (key 0 / candidate distance -1).  If every point is skipped the result is index 0, as in the
// reference (besti initialised to 0).
// =========================================================================================================
// max reductions for the sampling rounds as single instructions (v_max_f32_dpp; the compiler's form of the same DPP step is a
// v_mov 0, the DPP move, a canonicalising v_max and the v_max: the rounds are bound by instructions issued per round)
template <int ROR>
__device__ __forceinline__ float fps_max_ror(float x) {
  float r;
  asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x), "n"(ROR));
  return r;
}
__device__ __forceinline__ float fps_vmax(float a, float b) {
  float r;
  asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
__device__ __forceinline__ float fps_row_max(float v) {  // every lane: the maximum of its 16-lane row
  v = fps_max_ror<8>(v);
  v = fps_max_ror<4>(v);
  v = fps_max_ror<2>(v);
  return fps_max_ror<1>(v);
}
__device__ __forceinline__ float fps_wave_max(float v) {
  typedef unsigned u2_ __attribute__((ext_vector_type(2)));
  v = fps_row_max(v);
  // swap of two copies: one result holds the even rows' (lower half's) value in both rows of a pair, the other the odd rows' (upper half's)
  u2_ r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = fps_vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
  r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fps_vmax(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float fps_uniform(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// Round 4: a round is one serial chain (barrier -> best of the wave slots -> distances -> wave reduction -> slot) that every wave
// executes, so the kernel's time is 196 x (instructions per round x waves per SIMD x 4 cycles + the LDS / barrier latencies).  Lane t
// owns the PPT CONSECUTIVE points t PPT .. t PPT + PPT - 1, so inside a wave and across the waves a lower lane / wave means a lower
// index: the wave's winner is the lowest lane that holds the wave maximum (one DPP / permlane max reduction, a ballot and four
// v_readlane instead of six ds_bpermute steps on a 64-bit key), the round's winner the lowest slot that holds the maximum of the slots
// (lane l reads slot l, a 16-lane DPP maximum, a ballot, v_readlane), and a slot carries the winner's coordinates, so the next round does
// not start with a second, dependent LDS read of the point.  Same selection rule: the largest running min-distance, ties to the
// smallest index (sampling.cpp:96-126).
template <int THREADS, int PPT>
__global__ __launch_bounds__(THREADS) void fps_reg_kernel(const float* __restrict__ xyz, int N, int m,
                                                         int* __restrict__ out) {
  extern __shared__ float smem[];  // [N*3] points, then wave slots
  constexpr int NW = THREADS / 64;
  static_assert(NW <= 16, "one slot per lane of a 16-lane row");
  float* sp = smem;
  float4* slots = reinterpret_cast<float4*>(smem + ((N * 3 + 3) & ~3));  // [2][NW] {d, x, y, z}
  int* slotk = reinterpret_cast<int*>(slots + 2 * NW);                    // [2][NW] index

  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const float* p = xyz + (size_t)b * N * 3;
  int* o = out + (size_t)b * m;

  for (int i = t; i < N * 3; i += THREADS) sp[i] = p[i];
  __syncthreads();

  float px[PPT], py[PPT], pz[PPT], td[PPT];
  bool live[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = t * PPT + i;
    if (k < N) {
      px[i] = sp[k * 3 + 0];
      py[i] = sp[k * 3 + 1];
      pz[i] = sp[k * 3 + 2];
      const float mag = px[i] * px[i] + py[i] * py[i] + pz[i] * pz[i];
      live[i] = !((double)mag <= 1e-3);
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      live[i] = false;
    }
    td[i] = FLT_MAX;
  }
  if (t == 0) o[0] = 0;
  const float x0 = fps_uniform(sp[0]), y0 = fps_uniform(sp[1]), z0 = fps_uniform(sp[2]);
  float x1 = x0, y1 = y0, z1 = z0;  // the last selected point (wave-uniform: scalar registers)
  for (int j = 1; j < m; ++j) {
    // this lane's candidate: the largest min-distance of its live points, the first on a tie; -1: no live point (selects, no branches)
    float bd = -1.0f, bx = 0.f, by = 0.f, bz = 0.f;
    int bk = 0;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float d2 = (td[i] < d) ? td[i] : d;
      td[i] = live[i] ? d2 : td[i];
      const bool take = live[i] & (d2 > bd);
      bd = take ? d2 : bd;
      bk = take ? t * PPT + i : bk;
      bx = take ? px[i] : bx;
      by = take ? py[i] : by;
      bz = take ? pz[i] : bz;
    }
    const float wm = fps_wave_max(bd);
    const int wl = __ffsll((long long)__ballot(bd == wm)) - 1;  // lowest lane with the wave maximum = lowest index
    float4* sl = slots + (j & 1) * NW;
    int* sk = slotk + (j & 1) * NW;
    if ((t & 63) == wl) {
      sl[t >> 6] = make_float4(bd, bx, by, bz);
      sk[t >> 6] = bk;
    }
    __syncthreads();
    const float4 sv = sl[t & (NW - 1)];  // lane l of every 16-lane row: slot l mod NW
    int kv = sk[t & (NW - 1)];
    asm volatile("" : "+v"(kv));  // (both slot reads issued together: the index is not read behind a branch on the maximum)
    const float best = fps_row_max(sv.x);
    const int rl = __ffsll((long long)__ballot(sv.x == best)) - 1;  // lowest slot with the maximum (lanes 0 .. NW-1 come first)
    const bool none = fps_uniform(best) < 0.f;  // every point skipped: index 0, as in the reference (besti initialised to 0)
    int rk = __builtin_amdgcn_readlane(kv, rl);
    unsigned rx = __builtin_amdgcn_readlane(__float_as_uint(sv.y), rl), ry = __builtin_amdgcn_readlane(__float_as_uint(sv.z), rl),
             rz = __builtin_amdgcn_readlane(__float_as_uint(sv.w), rl);
    asm volatile("" : "+s"(rk), "+s"(rx), "+s"(ry), "+s"(rz));
    const int last = none ? 0 : rk;
    x1 = none ? x0 : __uint_as_float(rx);
    y1 = none ? y0 : __uint_as_float(ry);
    z1 = none ? z0 : __uint_as_float(rz);
    if (t == 0) o[j] = last;
  }
}

// Large-N variant (e.g. the 210 000-point template cloud of get_obj_feats, PEM/model/feature_extraction.py:152-158):
// running min-distances live in a global scratch row (`temp`, the same (B,N) buffer the reference allocates and never
// uses on CPU, sampling.cpp:192-194); points are re-read from global/L2 each round.  Same selection rule.
// `gate` != 0: the launch is the fallback behind fps_grid_kernel -- cloud b is recomputed only if that kernel raised its abort word
// (the first 64-bit word of fps_grid_ws(temp, b, N), inside this cloud's own scratch row, so no other workgroup can overwrite it).
__device__ __forceinline__ unsigned long long* fps_grid_ws(float* temp, int b, int N) {
  return reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(temp + (size_t)b * N) + 7) & ~(uintptr_t)7);
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void fps_big_kernel(const float* __restrict__ xyz, int N, int m,
                                                         float* __restrict__ temp, int* __restrict__ out, int gate) {
  constexpr int NW = THREADS / 64;
  __shared__ unsigned long long slots[2][NW];
  __shared__ unsigned long long s_gate;
  const int b = blockIdx.x, t = threadIdx.x;
  const float* p = xyz + (size_t)b * N * 3;
  float* td = temp + (size_t)b * N;
  int* o = out + (size_t)b * m;
  if (gate) {
    if (t == 0) s_gate = __hip_atomic_load(fps_grid_ws(temp, b, N), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_gate == 0ull) return;  // the grid kernel finished this cloud
  }
  for (int k = t; k < N; k += THREADS) td[k] = FLT_MAX;
  if (t == 0) o[0] = 0;
  int last = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = p[last * 3 + 0], y1 = p[last * 3 + 1], z1 = p[last * 3 + 2];
    unsigned long long key = 0ull;
    for (int k = t; k < N; k += THREADS) {
      const float x2 = p[k * 3 + 0], y2 = p[k * 3 + 1], z2 = p[k * 3 + 2];
      const float mag = x2 * x2 + y2 * y2 + z2 * z2;
      if ((double)mag <= 1e-3) continue;
      const float dx = x2 - x1, dy = y2 - y1, dz = z2 - z1;
      const float d = dx * dx + dy * dy + dz * dz;
      const float old = td[k];
      const float d2 = (old < d) ? old : d;
      td[k] = d2;
      const unsigned long long kk =
          ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(~(unsigned int)k);
      key = kk > key ? kk : key;
    }
    key = wave_max_u64(key);
    if ((t & 63) == 0) slots[j & 1][t >> 6] = key;
    __syncthreads();
    unsigned long long best = slots[j & 1][0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      const unsigned long long v = slots[j & 1][w];
      best = v > best ? v : best;
    }
    last = (best == 0ull) ? 0 : (int)(~(unsigned int)(best & 0xffffffffull));
    if (t == 0) o[j] = last;
  }
}

// Multi-workgroup variant for large clouds (the 210 000-point template cloud: 206 workgroups instead of one).
// Every workgroup keeps 256 x PPT points and their running min-distances in registers, so a round costs one distance
// update per point and no memory traffic beyond the grid-wide arg-max: the workgroup's best key goes into the round's
// 64-bit slot with a device-scope atomic max, a device-scope arrival counter closes the round, and every workgroup reads
// the winner back.  Same packed key and selection rule as above, so the result is identical to the one-workgroup kernels.
// ws per cloud: m round slots + arrival counter + abort flag (64-bit each), zeroed by the host before the launch.
// Co-residency: the host launches at most FPS_GRID_MAX_WG workgroups (one 256-thread workgroup per CU always fits
// beside whatever else is running, and nothing ever waits on this kernel), and the spin has a bounded exit: a workgroup
// that waits longer than the spin cap raises the cloud's abort word and every workgroup of that cloud leaves; the host always
// queues the one-workgroup kernel behind this one, gated on that word, so an aborted cloud is recomputed (slowly) instead of
// being returned truncated.  The grid is also checked against the occupancy API before this path is chosen.
#define FPS_GRID_MAX_WG 256
#define FPS_SPIN_CAP (1u << 24)
template <int PPT>
__global__ __launch_bounds__(256) void fps_grid_kernel(const float* __restrict__ xyz, int N, int m, float* __restrict__ temp,
                                                      int* __restrict__ out, unsigned spin_cap) {
  __shared__ unsigned long long slots[2][4];
  __shared__ unsigned long long s_best;
  const int g = blockIdx.x, G = gridDim.x, b = blockIdx.y, t = threadIdx.x;
  const float* p = xyz + (size_t)b * N * 3;
  int* o = out + (size_t)b * m;
  unsigned long long* abort_flag = fps_grid_ws(temp, b, N);  // [abort | arrived | best[0 .. m)]: zeroed by fps_grid_zero_kernel
  unsigned long long* arrived = abort_flag + 1;
  unsigned long long* best = abort_flag + 2;
  float px[PPT], py[PPT], pz[PPT], td[PPT];
  bool live[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int k = g * (256 * PPT) + i * 256 + t;
    if (k < N) {
      px[i] = p[(size_t)k * 3 + 0];
      py[i] = p[(size_t)k * 3 + 1];
      pz[i] = p[(size_t)k * 3 + 2];
      const float mag = px[i] * px[i] + py[i] * py[i] + pz[i] * pz[i];
      live[i] = !((double)mag <= 1e-3);
    } else {
      px[i] = py[i] = pz[i] = 0.f;
      live[i] = false;
    }
    td[i] = FLT_MAX;
  }
  if (g == 0 && t == 0) o[0] = 0;
  int last = 0;
  for (int j = 1; j < m; ++j) {
    const float x1 = p[(size_t)last * 3 + 0], y1 = p[(size_t)last * 3 + 1], z1 = p[(size_t)last * 3 + 2];
    unsigned long long key = 0ull;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      if (live[i]) {
        const float dx = px[i] - x1, dy = py[i] - y1, dz = pz[i] - z1;
        const float d = dx * dx + dy * dy + dz * dz;
        const float d2 = (td[i] < d) ? td[i] : d;
        td[i] = d2;
        const unsigned int k = (unsigned int)(g * (256 * PPT) + i * 256 + t);
        const unsigned long long kk = ((unsigned long long)__float_as_uint(d2) << 32) | (unsigned long long)(~k);
        key = kk > key ? kk : key;
      }
    }
    key = wave_max_u64(key);
    if ((t & 63) == 0) slots[j & 1][t >> 6] = key;
    __syncthreads();
    if (t == 0) {
      unsigned long long mine = slots[j & 1][0];
#pragma unroll
      for (int w = 1; w < 4; ++w) mine = slots[j & 1][w] > mine ? slots[j & 1][w] : mine;
      if (mine) __hip_atomic_fetch_max(&best[j], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(arrived, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long target = (unsigned long long)j * (unsigned long long)G;
      unsigned int polls = 0;
      bool dead = false;
      while (__hip_atomic_load(arrived, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull || ++polls > spin_cap) {
          __hip_atomic_store(abort_flag, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          dead = true;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      s_best = dead ? ~0ull : __hip_atomic_load(&best[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned long long win = s_best;
    if (win == ~0ull) return;  // aborted (~0 is no valid key: the distance field would be a NaN pattern); the gated fps_big_kernel
                               // launched right behind this kernel recomputes the cloud
    last = (win == 0ull) ? 0 : (int)(~(unsigned int)(win & 0xffffffffull));
    if (g == 0 && t == 0) o[j] = last;
  }
}

__global__ __launch_bounds__(256) void fps_grid_zero_kernel(float* __restrict__ temp, int N, int m) {
  unsigned long long* w = fps_grid_ws(temp, blockIdx.x, N);
  for (int i = threadIdx.x; i < m + 2; i += 256) w[i] = 0ull;
}

static unsigned g_fps_spin_cap = FPS_SPIN_CAP;
// test hook: polls a workgroup of the multi-workgroup FPS waits for the others before it gives the cloud up (0 = at once)
extern "C" int sam6d_fps_debug_spin_cap(long cap) {
  g_fps_spin_cap = cap < 0 ? FPS_SPIN_CAP : (unsigned)cap;
  return 0;
}

extern "C" int sam6d_furthest_point_sampling(const float* xyz, int B, int N, int m, float* temp, int* idx,
                                             void* stream) {
  SAM6D_REQUIRE(xyz && idx, "furthest_point_sampling: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && m >= 0, "furthest_point_sampling: bad sizes B=%d N=%d m=%d", B, N, m);
  if (B == 0 || m == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)((N * 3 + 3) & ~3) * 4 + 2 * 16 * (16 + 4);
  if (N <= 512 * 4) {  // (round 4, 64 clouds x 2048 -> 196: 108 us; 256 threads x 8 points per lane -- one wave per SIMD -- 144 us; round 3's
                       //  butterfly on a 64-bit key 161 us)
    hipLaunchKernelGGL((fps_reg_kernel<512, 4>), dim3(B), dim3(512), lds, s, xyz, N, m, idx);
  } else if (N <= 1024 * 4) {  // 48 KB of LDS points: stays under the 64 KB dynamic-LDS default
    hipLaunchKernelGGL((fps_reg_kernel<1024, 4>), dim3(B), dim3(1024), lds, s, xyz, N, m, idx);
  } else {
    SAM6D_REQUIRE(temp, "furthest_point_sampling: N=%d > 4096 needs the (B,N) float scratch `temp`", N);
    const long G = ((long)N + 1023) / 1024;  // 256 threads x 4 points per workgroup
    // co-residency of the hand-rolled grid barrier: all G * B workgroups must fit on the chip at once
    static int resident_dev[SAM6D_MAX_DEVICES];
    static unsigned long long resident_done = 0;
    int dev = 0;
    if (sam6d_first_use_on_device(&resident_done, &dev)) {
      SAM6D_REQUIRE(dev >= 0, "furthest_point_sampling: device ordinal beyond SAM6D_MAX_DEVICES");
      int cu = 0, per_cu = 0;
      hipError_t e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
      if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fps_grid_kernel<4>, 256, 0);
      resident_dev[dev] = (e == hipSuccess && cu > 0 && per_cu > 0) ? cu * (per_cu > 1 ? per_cu - 1 : 1) : 0;  // (margin: the API can be one high)
      (void)hipGetLastError();
      if (e == hipSuccess) sam6d_setup_done_on_device(&resident_done);  // (a failed query: one-workgroup kernel now, asked again next call)
    }
    const int max_resident = resident_dev[dev];
    const long cap_wg = max_resident < FPS_GRID_MAX_WG ? max_resident : FPS_GRID_MAX_WG;
    const bool grid_ok = G * B <= cap_wg && (size_t)(m + 2) * 8 + 8 <= (size_t)N * 4;  // the round slots live in the cloud's own row
    if (grid_ok) {
      hipLaunchKernelGGL(fps_grid_zero_kernel, dim3(B), dim3(256), 0, s, temp, N, m);
      hipLaunchKernelGGL((fps_grid_kernel<4>), dim3((unsigned)G, B), dim3(256), 0, s, xyz, N, m, temp, idx, g_fps_spin_cap);
      hipLaunchKernelGGL((fps_big_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, N, m, temp, idx, 1);  // runs only for aborted clouds
    } else {
      hipLaunchKernelGGL((fps_big_kernel<1024>), dim3(B), dim3(1024), 0, s, xyz, N, m, temp, idx, 0);
    }
  }
  SAM6D_LAUNCH_CHECK("furthest_point_sampling");
}

// =========================================================================================================
// gather_points: out[b,c,j] = points[b,c,idx[b,j]], 0 for an out-of-range index (sampling.cpp:23-44).
// (B,C,N) channel-major layout as the `_ext` seam has it; the pipeline itself uses gather_rows below.
// =========================================================================================================
__global__ void gather_points_kernel(const float* __restrict__ points, const int* __restrict__ idx, int C, int N, int M,
                                     float* __restrict__ out) {
  const int b = blockIdx.z;
  const int c = blockIdx.y;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= M) return;
  const int a = idx[(size_t)b * M + j];
  const float* row = points + ((size_t)b * C + c) * N;
  out[((size_t)b * C + c) * M + j] = (a >= 0 && a < N) ? row[a] : 0.0f;
}

extern "C" int sam6d_gather_points(const float* points, const int* idx, int B, int C, int N, int M, float* out,
                                   void* stream) {
  SAM6D_REQUIRE(points && idx && out, "gather_points: null pointer");
  SAM6D_REQUIRE(B >= 0 && C >= 0 && N > 0 && M >= 0, "gather_points: bad sizes");
  SAM6D_REQUIRE(C <= 65535 && B <= 65535, "gather_points: C and B must be <= 65535");
  if (B == 0 || C == 0 || M == 0) return 0;
  dim3 grid(cdiv(M, 256), C, B);
  hipLaunchKernelGGL(gather_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, points, idx, C, N, M, out);
  SAM6D_LAUNCH_CHECK("gather_points");
}

// Row gather on the (B,N,C) layout the pipeline keeps its features in: out[b,j,:] = feats[b,idx[b,j]+off,:].
// One wave per output row, 16-byte lanes when C % 4 == 0.  Out-of-range index -> zeros (same rule as gather_points).
__global__ void gather_rows_kernel(const float* __restrict__ feats, const int* __restrict__ idx, int N, int M, int C,
                                   long in_stride_b, long out_stride_b, int idx_off, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (j >= M) return;
  const int a = idx[(size_t)b * M + j] + idx_off;
  const bool ok = (a >= 0 && a < N);
  const float* srow = feats + (size_t)b * in_stride_b + (size_t)(ok ? a : 0) * C;
  float* drow = out + (size_t)b * out_stride_b + (size_t)j * C;
  if ((C & 3) == 0) {
    const float4* src = reinterpret_cast<const float4*>(srow);
    float4* dst = reinterpret_cast<float4*>(drow);
    for (int c = (threadIdx.x & 63); c < C / 4; c += 64) dst[c] = ok ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
  } else {
    for (int c = (threadIdx.x & 63); c < C; c += 64) drow[c] = ok ? srow[c] : 0.f;
  }
}

extern "C" int sam6d_gather_rows(const float* feats, const int* idx, int B, int N, int M, int C, long in_stride_b,
                                 long out_stride_b, int idx_off, float* out, void* stream) {
  SAM6D_REQUIRE(feats && idx && out, "gather_rows: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0 && B <= 65535, "gather_rows: bad sizes");
  SAM6D_REQUIRE((C & 3) != 0 || (((in_stride_b | out_stride_b) & 3) == 0 && (((size_t)feats | (size_t)out) & 15) == 0),
                "gather_rows: 16-byte alignment required when C %% 4 == 0");
  if (B == 0 || M == 0) return 0;
  dim3 grid(cdiv(M, 4), B);
  hipLaunchKernelGGL(gather_rows_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, idx, N, M, C,
                     in_stride_b, out_stride_b, idx_off, out);
  SAM6D_LAUNCH_CHECK("gather_rows");
}

// gather_rows with row 0 of every feats[b] supplied separately: `lead[b]` stands for feats[b, 0, :] (which need not be written), and
// out[b,0,:] = lead[b,:], out[b,1+j,:] = feats[b, idx[b,j] + idx_off, :] -- the sparse tokens of a sparse-to-dense block (bg token + the
// FPS rows of the dense tokens, PEM/model/transformer.py:667-705) in one launch instead of a gather and a one-row copy.
__global__ void gather_rows_lead_kernel(const float* __restrict__ feats, const int* __restrict__ idx, int N, int M, int C,
                                        long in_stride_b, long out_stride_b, int idx_off, const float* __restrict__ lead,
                                        long lead_stride_b, float* __restrict__ out) {
  const int b = blockIdx.y;
  const int j = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);  // output row 0 .. M
  if (j > M) return;
  bool ok = true;
  const float* srow;
  if (j == 0) {
    srow = lead + (size_t)b * lead_stride_b;
  } else {
    const int a = idx[(size_t)b * M + j - 1] + idx_off;
    ok = (a >= 0 && a < N);
    // row 0 of feats[b] IS the lead row (the caller may not have written it into feats): the FPS indices address the cat [bg; dense]
    // without the + 1 (the reference's quirk, transformer.py:667-705), and FPS always starts at index 0
    srow = (ok && a > 0) ? feats + (size_t)b * in_stride_b + (size_t)a * C : lead + (size_t)b * lead_stride_b;
  }
  float* drow = out + (size_t)b * out_stride_b + (size_t)j * C;
  const float4* src = reinterpret_cast<const float4*>(srow);
  float4* dst = reinterpret_cast<float4*>(drow);
  for (int c = (threadIdx.x & 63); c < C / 4; c += 64) dst[c] = ok ? src[c] : make_float4(0.f, 0.f, 0.f, 0.f);
}

extern "C" int sam6d_gather_rows_lead(const float* feats, const int* idx, int B, int N, int M, int C, long in_stride_b,
                                      long out_stride_b, int idx_off, const float* lead, long lead_stride_b, float* out,
                                      void* stream) {
  SAM6D_REQUIRE(feats && idx && out && lead, "gather_rows_lead: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && M >= 0 && C > 0 && (C & 3) == 0 && B <= 65535, "gather_rows_lead: bad sizes (C %% 4 == 0)");
  SAM6D_REQUIRE(((in_stride_b | out_stride_b | lead_stride_b) & 3) == 0 && (((size_t)feats | (size_t)out | (size_t)lead) & 15) == 0,
                "gather_rows_lead: 16-byte alignment required");
  if (B == 0) return 0;
  dim3 grid(cdiv(M + 1, 4), B);
  hipLaunchKernelGGL(gather_rows_lead_kernel, grid, dim3(256), 0, (hipStream_t)stream, feats, idx, N, M, C, in_stride_b, out_stride_b,
                     idx_off, lead, lead_stride_b, out);
  SAM6D_LAUNCH_CHECK("gather_rows_lead");
}

// =========================================================================================================
// Ball query (ball_query.cpp:16-62): per query, the first `nsample` indices k (increasing) with d2 < r*r; the first
// hit pre-fills every slot; no hit -> zeros.  d2 in plain fp32, source order, no FMA.
// One wave per query: 64 candidates per step from an LDS-staged chunk of the cloud, ballot + prefix popcount keep
// the reference's index order.  QPB queries per block, candidate chunks of CH points (24 KB: the 1024 workgroups of a 32 x 2048 query set are resident at once) so any N works.
// =========================================================================================================
#define BQ_CH 2048
#define BQ_QPB 64
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz, const float* __restrict__ xyz,
                                                         int N, int M, float radius2, int nsample,
                                                         int* __restrict__ idx) {
  __shared__ float sp[BQ_CH * 3];
  __shared__ int s_cnt[BQ_QPB];
  __shared__ int s_first[BQ_QPB];
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * BQ_QPB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cx = xyz + (size_t)b * N * 3;
  const float* cq = new_xyz + (size_t)b * M * 3;
  int* ci = idx + (size_t)b * M * nsample;
  if (threadIdx.x < BQ_QPB) {
    s_cnt[threadIdx.x] = 0;
    s_first[threadIdx.x] = 0;
  }
  for (int base = 0; base < N; base += BQ_CH) {
    const int cn = min(BQ_CH, N - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cn * 3; i += 256) sp[i] = cx[(size_t)base * 3 + i];
    __syncthreads();
    for (int ql = wave; ql < BQ_QPB; ql += 4) {
      const int q = q0 + ql;
      if (q >= M) break;
      int cnt = s_cnt[ql];
      if (cnt >= nsample) continue;
      const float qx = cq[q * 3 + 0], qy = cq[q * 3 + 1], qz = cq[q * 3 + 2];
      int first = s_first[ql];
      for (int k0 = 0; k0 < cn && cnt < nsample; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < cn) {
          const float x = sp[k * 3 + 0], y = sp[k * 3 + 1], z = sp[k * 3 + 2];
          const float d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
          hit = d2 < radius2;
        }
        const unsigned long long mask = __ballot(hit);
        if (mask) {
          if (cnt == 0) first = base + k0 + (__ffsll((long long)mask) - 1);
          const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
          if (hit && pos < nsample) ci[(size_t)q * nsample + pos] = base + k;
          cnt += __popcll(mask);
        }
      }
      if (lane == 0) {
        s_cnt[ql] = cnt;
        s_first[ql] = first;
      }
    }
  }
  __syncthreads();
  // tail fill: slots [cnt, nsample) hold the first hit (or 0 when the ball is empty)
  for (int ql = wave; ql < BQ_QPB; ql += 4) {
    const int q = q0 + ql;
    if (q >= M) break;
    const int cnt = min(s_cnt[ql], nsample), first = s_first[ql];
    for (int l = cnt + lane; l < nsample; l += 64) ci[(size_t)q * nsample + l] = first;
  }
}

// Two radii in one pass over the candidates (PositionalEncoding queries the same cloud with r1 < r2: fine_point_matching.py
// :108-131): the distance of a (query, candidate) pair is computed once, each radius keeps its own hit counter and index list.
// Same per-radius semantics as ball_query_kernel (index order, first hit pre-fills, empty ball -> zeros).
__global__ __launch_bounds__(256) void ball_query2_kernel(const float* __restrict__ new_xyz, const float* __restrict__ xyz, int N,
                                                          int M, float r2a, int nsa, int* __restrict__ idxa, float r2b, int nsb,
                                                          int* __restrict__ idxb) {
  __shared__ float sp[BQ_CH * 3];
  __shared__ int s_cnt[2][BQ_QPB];
  __shared__ int s_first[2][BQ_QPB];
  const int b = blockIdx.y;
  const int q0 = blockIdx.x * BQ_QPB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* cx = xyz + (size_t)b * N * 3;
  const float* cq = new_xyz + (size_t)b * M * 3;
  int* cia = idxa + (size_t)b * M * nsa;
  int* cib = idxb + (size_t)b * M * nsb;
  if (threadIdx.x < BQ_QPB) {
    s_cnt[0][threadIdx.x] = s_cnt[1][threadIdx.x] = 0;
    s_first[0][threadIdx.x] = s_first[1][threadIdx.x] = 0;
  }
  for (int base = 0; base < N; base += BQ_CH) {
    const int cn = min(BQ_CH, N - base);
    __syncthreads();
    for (int i = threadIdx.x; i < cn * 3; i += 256) sp[i] = cx[(size_t)base * 3 + i];
    __syncthreads();
    for (int ql = wave; ql < BQ_QPB; ql += 4) {
      const int q = q0 + ql;
      if (q >= M) break;
      int ca = s_cnt[0][ql], cb = s_cnt[1][ql];
      if (ca >= nsa && cb >= nsb) continue;
      const float qx = cq[q * 3 + 0], qy = cq[q * 3 + 1], qz = cq[q * 3 + 2];
      int fa = s_first[0][ql], fb = s_first[1][ql];
      for (int k0 = 0; k0 < cn && (ca < nsa || cb < nsb); k0 += 64) {
        const int k = k0 + lane;
        float d2 = INFINITY;
        if (k < cn) {
          const float x = sp[k * 3 + 0], y = sp[k * 3 + 1], z = sp[k * 3 + 2];
          d2 = (qx - x) * (qx - x) + (qy - y) * (qy - y) + (qz - z) * (qz - z);
        }
        if (ca < nsa) {
          const bool hit = d2 < r2a;
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            if (ca == 0) fa = base + k0 + (__ffsll((long long)mask) - 1);
            const int pos = ca + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < nsa) cia[(size_t)q * nsa + pos] = base + k;
            ca += __popcll(mask);
          }
        }
        if (cb < nsb) {
          const bool hit = d2 < r2b;
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            if (cb == 0) fb = base + k0 + (__ffsll((long long)mask) - 1);
            const int pos = cb + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < nsb) cib[(size_t)q * nsb + pos] = base + k;
            cb += __popcll(mask);
          }
        }
      }
      if (lane == 0) {
        s_cnt[0][ql] = ca;
        s_cnt[1][ql] = cb;
        s_first[0][ql] = fa;
        s_first[1][ql] = fb;
      }
    }
  }
  __syncthreads();
  for (int ql = wave; ql < BQ_QPB; ql += 4) {  // tail fill: slots [cnt, nsample) hold the first hit (or 0 when the ball is empty)
    const int q = q0 + ql;
    if (q >= M) break;
    const int ca = min(s_cnt[0][ql], nsa), cb = min(s_cnt[1][ql], nsb), fa = s_first[0][ql], fb = s_first[1][ql];
    for (int l = ca + lane; l < nsa; l += 64) cia[(size_t)q * nsa + l] = fa;
    for (int l = cb + lane; l < nsb; l += 64) cib[(size_t)q * nsb + l] = fb;
  }
}

extern "C" int sam6d_ball_query2(const float* new_xyz, const float* xyz, int B, int N, int M, float radius1, int nsample1, int* idx1,
                                 float radius2, int nsample2, int* idx2, void* stream) {
  SAM6D_REQUIRE(new_xyz && xyz && idx1 && idx2, "ball_query2: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample1 > 0 && nsample2 > 0, "ball_query2: bad sizes");
  SAM6D_REQUIRE(B <= 65535, "ball_query2: B must be <= 65535");
  if (B == 0 || M == 0) return 0;
  dim3 grid(cdiv(M, BQ_QPB), B);
  hipLaunchKernelGGL(ball_query2_kernel, grid, dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, N, M, radius1 * radius1, nsample1,
                     idx1, radius2 * radius2, nsample2, idx2);
  SAM6D_LAUNCH_CHECK("ball_query2");
}

// ---------------------------------------------------------------------------------------------------------
// The two-radius ball query with spatial pruning -- same outputs, bit for bit (ball_query.cpp:16-62).
// The all-pairs scan tests 2048 candidates per query to find ~70 hits.  Here the cloud is bucketed once into a G^3 grid of cells of
// edge c = 1.001 max(r1, r2) (origin = the cloud's min corner, cell coordinates clamped to [0, G)): any point the fp32 test
// d2 < r^2 accepts lies within +-1 cell of the query's cell in every axis (the 0.1 % margin covers the rounding of both the test and
// the cell coordinates; clamping only merges cells).  A wave owns a query and scans the 9 (dy, dz) runs of up to 3 x-adjacent cells
// (contiguous in the cell-sorted point list), 64 candidates per step: ~450 candidates instead of 2048 for a uniform unit cube.
// The reference's result is the first `nsample` hits IN INDEX ORDER, and the grid visits candidates in cell order -- so hits only set
// bit `index` of a per-query bit mask in LDS (ds_or_b32), and the ordered output is read off the mask afterwards: a prefix sum of the
// words' popcounts over the wave gives every set bit its output slot.  The order inside a cell (the scatter uses LDS atomics) never
// matters.  Workspace per cloud: N float4 (x, y, z, index) sorted by cell | G^3 + 1 cell starts | origin, 1 / c.
// ---------------------------------------------------------------------------------------------------------
#define BQG_G 16
#define BQG_CELLS (BQG_G * BQG_G * BQG_G)
#define BQG_MAXN 8192
#define BQG_QPW 4            // queries per wave
__host__ __device__ inline size_t bqg_cloud_bytes(int N) { return (size_t)N * 16 + (size_t)(BQG_CELLS + 1) * 4 + 12 + 16; }

// inclusive prefix sum over the 64 lanes on the DPP path (no ds_bpermute round trips): Hillis-Steele inside each 16-lane row (row_shr
// 1, 2, 4, 8 with zero fill), then the row totals of the rows before (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3)
__device__ __forceinline__ int wave_incl_scan_i32_dpp(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false); // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false); // row_bcast:31 -> rows 2, 3
  return v;
}
__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ int bqg_cell1(float v, float o, float inv_c) {
  const float u = fminf(fmaxf((v - o) * inv_c, 0.0f), (float)(BQG_G - 1));  // (NaN -> 0)
  return (int)u;
}

__global__ __launch_bounds__(1024) void bqg_build_kernel(const float* __restrict__ xyz, int N, float cell, unsigned char* __restrict__ ws) {
  __shared__ int hist[BQG_CELLS];
  __shared__ float red[3][16];
  __shared__ int wsum[16];
  const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const float* p = xyz + (size_t)b * N * 3;
  unsigned char* wb = ws + (size_t)b * ((bqg_cloud_bytes(N) + 15) & ~(size_t)15);
  float4* sorted = reinterpret_cast<float4*>(wb);
  int* start = reinterpret_cast<int*>(wb + (size_t)N * 16);
  float* org = reinterpret_cast<float*>(wb + (size_t)N * 16 + (size_t)(BQG_CELLS + 1) * 4);
  float mx = INFINITY, my = INFINITY, mz = INFINITY;
  for (int i = t; i < N; i += 1024) {
    const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    mx = fminf(mx, x); my = fminf(my, y); mz = fminf(mz, z);  // (fminf ignores NaN)
  }
  mx = wave_min(mx); my = wave_min(my); mz = wave_min(mz);
  if (lane == 0) { red[0][wave] = mx; red[1][wave] = my; red[2][wave] = mz; }
  for (int i = t; i < BQG_CELLS; i += 1024) hist[i] = 0;
  __syncthreads();
  float ox = red[0][0], oy = red[1][0], oz = red[2][0];
#pragma unroll
  for (int w = 1; w < 16; ++w) { ox = fminf(ox, red[0][w]); oy = fminf(oy, red[1][w]); oz = fminf(oz, red[2][w]); }
  if (!(ox > -3.0e38f && ox < 3.0e38f)) ox = 0.f;
  if (!(oy > -3.0e38f && oy < 3.0e38f)) oy = 0.f;
  if (!(oz > -3.0e38f && oz < 3.0e38f)) oz = 0.f;
  const float inv_c = 1.0f / cell;
  if (t == 0) { org[0] = ox; org[1] = oy; org[2] = oz; org[3] = inv_c; }
  for (int i = t; i < N; i += 1024) {
    const int c = (bqg_cell1(p[i * 3 + 2], oz, inv_c) * BQG_G + bqg_cell1(p[i * 3 + 1], oy, inv_c)) * BQG_G + bqg_cell1(p[i * 3], ox, inv_c);
    atomicAdd(&hist[c], 1);
  }
  __syncthreads();
  // exclusive scan of the 4096 counts: 4 per thread, wave scan, wave totals
  int v[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) { v[u] = hist[t * 4 + u]; sum += v[u]; }
  int inc = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int nb = __shfl_up(inc, o, 64);
    if (lane >= o) inc += nb;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  int run = base + inc - sum;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    start[t * 4 + u] = run;
    hist[t * 4 + u] = run;   // becomes the scatter cursor
    run += v[u];
  }
  if (t == 1023) start[BQG_CELLS] = run;
  __syncthreads();
  for (int i = t; i < N; i += 1024) {
    const float x = p[i * 3], y = p[i * 3 + 1], z = p[i * 3 + 2];
    const int c = (bqg_cell1(z, oz, inv_c) * BQG_G + bqg_cell1(y, oy, inv_c)) * BQG_G + bqg_cell1(x, ox, inv_c);
    const int slot = atomicAdd(&hist[c], 1);
    sorted[slot] = make_float4(x, y, z, __int_as_float(i));
  }
}

template <int WPL>  // mask words per lane: N <= 2048 * WPL
__global__ __launch_bounds__(256) void bqg_query_kernel(const float* __restrict__ new_xyz, int N, int M, float r2a, int nsa,
                                                        int* __restrict__ idxa, float r2b, int nsb, int* __restrict__ idxb,
                                                        const unsigned char* __restrict__ ws) {
  __shared__ unsigned maskA[4][64 * WPL];
  __shared__ unsigned maskB[4][64 * WPL];
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned char* wb = ws + (size_t)b * ((bqg_cloud_bytes(N) + 15) & ~(size_t)15);
  const float4* sorted = reinterpret_cast<const float4*>(wb);
  const int* start = reinterpret_cast<const int*>(wb + (size_t)N * 16);
  const float* org = reinterpret_cast<const float*>(wb + (size_t)N * 16 + (size_t)(BQG_CELLS + 1) * 4);
  const float ox = org[0], oy = org[1], oz = org[2], inv_c = org[3];
  unsigned* mA = maskA[wave];
  unsigned* mB = maskB[wave];
  for (int qi = 0; qi < BQG_QPW; ++qi) {
    const int q = (blockIdx.x * 4 + wave) * BQG_QPW + qi;
    if (q >= M) break;  // (wave-uniform)
    const float* qp = new_xyz + ((size_t)b * M + q) * 3;
    const float qx = qp[0], qy = qp[1], qz = qp[2];
#pragma unroll
    for (int j = 0; j < WPL; ++j) { mA[lane * WPL + j] = 0u; mB[lane * WPL + j] = 0u; }
    const int cx = bqg_cell1(qx, ox, inv_c), cy = bqg_cell1(qy, oy, inv_c), cz = bqg_cell1(qz, oz, inv_c);
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, BQG_G - 1);
    for (int dz = -1; dz <= 1; ++dz) {
      const int z = cz + dz;
      if (z < 0 || z >= BQG_G) continue;
      for (int dy = -1; dy <= 1; ++dy) {
        const int y = cy + dy;
        if (y < 0 || y >= BQG_G) continue;
        const int row = (z * BQG_G + y) * BQG_G;
        const int lo = start[row + x0], hi = start[row + x1 + 1];
        for (int k = lo + lane; k < hi; k += 64) {
          const float4 c = sorted[k];
          const float d2 = (qx - c.x) * (qx - c.x) + (qy - c.y) * (qy - c.y) + (qz - c.z) * (qz - c.z);
          const int id = __float_as_int(c.w);
          if (d2 < r2b) atomicOr(&mB[id >> 5], 1u << (id & 31));
          if (d2 < r2a) atomicOr(&mA[id >> 5], 1u << (id & 31));
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the wave's LDS atomics are ordered before the reads below
    // ordered read-out of a mask: lane L owns words L*WPL .. L*WPL+WPL-1 (ascending indices)
    auto emit = [&](const unsigned* m, int ns, int* out) {
      unsigned w[WPL];
      int cnt = 0;
#pragma unroll
      for (int j = 0; j < WPL; ++j) { w[j] = m[lane * WPL + j]; cnt += __popc(w[j]); }
      const int inc = wave_incl_scan_i32_dpp(cnt);
      const int total = __builtin_amdgcn_readlane(inc, 63);
      int pos = inc - cnt;
      // the first hit: lowest set bit of the first non-empty lane (lane order = index order)
      int mine = 0;
#pragma unroll
      for (int j = WPL - 1; j >= 0; --j)
        if (w[j]) mine = 32 * (lane * WPL + j) + (__ffs(w[j]) - 1);
      const unsigned long long any = __ballot(cnt > 0);
      const int first = any ? __builtin_amdgcn_readlane(mine, __ffsll((long long)any) - 1) : 0;
#pragma unroll
      for (int j = 0; j < WPL; ++j) {
        unsigned x = w[j];
        while (x && pos < ns) {
          out[pos++] = 32 * (lane * WPL + j) + (__ffs(x) - 1);
          x &= x - 1;
        }
      }
      for (int l = min(total, ns) + lane; l < ns; l += 64) out[l] = first;
    };
    emit(mA, nsa, idxa + ((size_t)b * M + q) * nsa);
    emit(mB, nsb, idxb + ((size_t)b * M + q) * nsb);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
}

extern "C" size_t sam6d_ball_query2_grid_workspace_bytes(int B, int N) {
  return (size_t)B * ((bqg_cloud_bytes(N) + 15) & ~(size_t)15);
}

extern "C" int sam6d_ball_query2_grid(const float* new_xyz, const float* xyz, int B, int N, int M, float radius1, int nsample1, int* idx1,
                                      float radius2, int nsample2, int* idx2, void* ws, size_t ws_bytes, void* stream) {
  SAM6D_REQUIRE(new_xyz && xyz && idx1 && idx2 && ws, "ball_query2_grid: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample1 > 0 && nsample2 > 0 && B <= 65535, "ball_query2_grid: bad sizes");
  SAM6D_REQUIRE((((size_t)ws) & 15) == 0 && ws_bytes >= sam6d_ball_query2_grid_workspace_bytes(B, N),
                "ball_query2_grid: workspace too small / not 16-byte aligned (sam6d_ball_query2_grid_workspace_bytes)");
  const float rmax = radius1 > radius2 ? radius1 : radius2;
  if (N > BQG_MAXN || !(rmax > 0.f) || !(rmax < 1.0e30f))  // outside the pruned kernel's range: the all-pairs scan, same results
    return sam6d_ball_query2(new_xyz, xyz, B, N, M, radius1, nsample1, idx1, radius2, nsample2, idx2, stream);
  if (B == 0 || M == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bqg_build_kernel, dim3(B), dim3(1024), 0, s, xyz, N, rmax * 1.001f, (unsigned char*)ws);
  SAM6D_LAUNCH_CHECK_CONT("ball_query2_grid(build)");
  const dim3 grid(cdiv(M, 4 * BQG_QPW), B);
  const float r2a = radius1 * radius1, r2b = radius2 * radius2;  // fp32 products, as the reference (ball_query.cpp:20)
  if (N <= 2048)
    hipLaunchKernelGGL((bqg_query_kernel<1>), grid, dim3(256), 0, s, new_xyz, N, M, r2a, nsample1, idx1, r2b, nsample2, idx2, (const unsigned char*)ws);
  else if (N <= 4096)
    hipLaunchKernelGGL((bqg_query_kernel<2>), grid, dim3(256), 0, s, new_xyz, N, M, r2a, nsample1, idx1, r2b, nsample2, idx2, (const unsigned char*)ws);
  else
    hipLaunchKernelGGL((bqg_query_kernel<4>), grid, dim3(256), 0, s, new_xyz, N, M, r2a, nsample1, idx1, r2b, nsample2, idx2, (const unsigned char*)ws);
  SAM6D_LAUNCH_CHECK("ball_query2_grid");
}

extern "C" int sam6d_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample,
                                int* idx, void* stream) {
  SAM6D_REQUIRE(new_xyz && xyz && idx, "ball_query: null pointer");
  SAM6D_REQUIRE(B >= 0 && N > 0 && M >= 0 && nsample > 0, "ball_query: bad sizes");
  SAM6D_REQUIRE(B <= 65535, "ball_query: B must be <= 65535");
  if (B == 0 || M == 0) return 0;
  const float r2 = radius * radius;  // fp32 product, as the reference (ball_query.cpp:20)
  dim3 grid(cdiv(M, BQ_QPB), B);
  hipLaunchKernelGGL(ball_query_kernel, grid, dim3(256), 0, (hipStream_t)stream, new_xyz, xyz, N, M, r2, nsample, idx);
  SAM6D_LAUNCH_CHECK("ball_query");
}

// =========================================================================================================
// group_points: out[b,c,j,k] = points[b,c,idx[b,j,k]] (0 when out of range), group_points.cpp:20-45.
// =========================================================================================================
__global__ void group_points_kernel(const float* __restrict__ points, const int* __restrict__ idx, int C, int N,
                                    long MS, float* __restrict__ out) {
  const int b = blockIdx.y;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= MS) return;
  const int a = idx[(size_t)b * MS + e];
  const bool ok = (a >= 0 && a < N);
  const float* pb = points + (size_t)b * C * N;
  float* ob = out + (size_t)b * C * MS;
  for (int c = 0; c < C; ++c) ob[(size_t)c * MS + e] = ok ? pb[(size_t)c * N + a] : 0.0f;
}

extern "C" int sam6d_group_points(const float* points, const int* idx, int B, int C, int N, int M, int S, float* out,
                                  void* stream) {
  SAM6D_REQUIRE(points && idx && out, "group_points: null pointer");
  SAM6D_REQUIRE(B >= 0 && C >= 0 && N > 0 && M >= 0 && S >= 0, "group_points: bad sizes");
  SAM6D_REQUIRE(B <= 65535, "group_points: B must be <= 65535");
  if (B == 0 || C == 0 || M == 0 || S == 0) return 0;
  const long MS = (long)M * S;
  dim3 grid((unsigned)((MS + 255) / 256), B);
  hipLaunchKernelGGL(group_points_kernel, grid, dim3(256), 0, (hipStream_t)stream, points, idx, C, N, MS, out);
  SAM6D_LAUNCH_CHECK("group_points");
}
