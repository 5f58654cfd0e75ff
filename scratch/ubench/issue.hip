// Micro-benchmark: cycles per v_mfma_f32_16x16x32_f16 with NV vector instructions of one kind in every gap (gfx950).
// build: hipcc --offload-arch=gfx950 -O3 issue.hip -o issue ; run: ./issue
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
__device__ __forceinline__ void filler(float& a, float& b, float& c, unsigned& u, float* lds) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));
  if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(u) : "v"(b), "v"(c));
  if (KIND == 3) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(a) : "v"(u), "v"(b));
  if (KIND == 4) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  if (KIND == 5) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
  if (KIND == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(a) : "v"(b));
}

template <int KIND, int NV>
__global__ __launch_bounds__(512) void k(unsigned long long* out, float seed) {
  __shared__ float lds[4096];
  const int lane = threadIdx.x & 63;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + lane * 0.01f + i); b[i] = (_Float16)(seed * 0.5f - lane * 0.02f + i); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8], y = seed, z = seed * 2;
  unsigned u = 0x3c003c00u;
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) filler<KIND>(x[(m * NV + v) & 7], y, z, u, lds);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) out[1000000] = 1;  // keep everything alive
  if (lane == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7>
__global__ __launch_bounds__(768) void kp(unsigned long long* out, float seed) {
  const int lane = threadIdx.x & 63;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + lane * 0.01f + i); b[i] = (_Float16)(seed * 0.5f - lane * 0.02f + i); }
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float x[8], y = seed, z = seed * 2;
  unsigned u = 0x3c003c00u;
  for (int i = 0; i < 8; ++i) x[i] = seed + i;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  constexpr int P[8] = {P0, P1, P2, P3, P4, P5, P6, P7};
  for (int it = 0; it < 256; ++it) {
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      acc[m & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[m & 3], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < P[m]; ++v) filler<KIND>(x[(m + v) & 7], y, z, u, nullptr);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  if (s == 12345.678f) out[1000000] = 1;
  if (lane == 0) {
    out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6))] = t0;
    out[2 * (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) + 1] = t1;
  }
}

template <int KIND, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7>
void runp(unsigned long long* d, int threads) {
  hipLaunchKernelGGL((kp<KIND, P0, P1, P2, P3, P4, P5, P6, P7>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
  hipLaunchKernelGGL((kp<KIND, P0, P1, P2, P3, P4, P5, P6, P7>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
  hipDeviceSynchronize();
  const int wpg = threads / 64, nw = 256 * wpg;
  unsigned long long* h = (unsigned long long*)malloc(nw * 16);
  hipMemcpy(h, d, nw * 16, hipMemcpyDeviceToHost);
  double span = 0, own = 0;
  for (int g = 0; g < 256; ++g) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < wpg; ++w) {
      const unsigned long long a = h[2 * (g * wpg + w)], b = h[2 * (g * wpg + w) + 1];
      lo = a < lo ? a : lo; hi = b > hi ? b : hi; own += (double)(b - a);
    }
    span += (double)(hi - lo);
  }
  const int wps = wpg / 4;
  printf("pattern %d%d%d%d%d%d%d%d (avg %.2f VALU per gap) waves/SIMD %d: %6.1f cycles per MFMA per SIMD (workgroup span), wave's own %6.1f\n", P0, P1, P2, P3,
         P4, P5, P6, P7, (P0 + P1 + P2 + P3 + P4 + P5 + P6 + P7) / 8.0, wps, span / 256 / (256 * 8) / wps, own / nw / (256 * 8));
  free(h);
}

template <int KIND, int NV>
void run(const char* name, unsigned long long* d, int threads) {
  hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(256), dim3(threads), 0, 0, d, 1.0f);
  hipDeviceSynchronize();
  const int nw = 256 * threads / 64;
  unsigned long long* h = (unsigned long long*)malloc(nw * 8);
  hipMemcpy(h, d, nw * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (int i = 0; i < nw; ++i) s += (double)h[i];
  printf("%-22s NV %d  waves/SIMD %d : %6.1f cycles per MFMA (wave), %6.1f per MFMA (SIMD)\n", name, NV, threads / 256, s / nw / (256 * 8),
         s / nw / (256 * 8) / (threads / 256));
  free(h);
}

#define ROW(KIND, name)                                   \
  run<KIND, 0>(name, d, th); run<KIND, 1>(name, d, th); run<KIND, 2>(name, d, th); run<KIND, 3>(name, d, th); run<KIND, 4>(name, d, th);

int main() {
  unsigned long long* d;
  hipMalloc(&d, 8 * 1000008);
  for (int th = 256; th <= 768; th += 256) {
    printf("-- v_fma_f32 fillers\n");
    runp<0, 0, 0, 0, 0, 0, 0, 0, 0>(d, th);
    runp<0, 1, 1, 1, 1, 1, 1, 1, 1>(d, th);
    runp<0, 2, 2, 2, 2, 2, 2, 2, 2>(d, th);
    runp<0, 1, 3, 1, 3, 1, 3, 1, 3>(d, th);
    runp<0, 3, 3, 3, 3, 3, 3, 3, 3>(d, th);
    runp<0, 2, 4, 2, 4, 2, 4, 2, 4>(d, th);
    runp<0, 1, 1, 1, 5, 1, 1, 1, 5>(d, th);
    runp<0, 0, 0, 0, 8, 0, 0, 0, 8>(d, th);
    runp<0, 0, 4, 0, 4, 0, 4, 0, 4>(d, th);
    runp<0, 1, 1, 3, 3, 1, 1, 3, 3>(d, th);
    printf("-- v_cvt_pk_f16_f32 fillers\n");
    runp<2, 2, 2, 2, 2, 2, 2, 2, 2>(d, th);
    runp<2, 1, 3, 1, 3, 1, 3, 1, 3>(d, th);
  }
  for (int th = 256; th <= 0; th += 256) {
    ROW(0, "v_fma_f32")
    ROW(1, "v_max3_f32")
    ROW(2, "v_cvt_pk_f16_f32")
    ROW(3, "v_fma_mix_f32")
    ROW(4, "v_permlane32_swap")
    ROW(5, "v_permlane16_swap")
    ROW(7, "v_mov_b32")
  }
  return 0;
}
