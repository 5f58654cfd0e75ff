// Is v_mfma_f32_32x32x2_f32 with a zero accumulator the chain fma(a1,b1, rn(a0*b0)), and a second one fma(a3,b3, fma(a2,b2,acc))?
// (the K = 3 pairwise-distance recipe: xy = fma(x2,y2, fma(x1,y1, x0*y0)).)  Prints the number of mismatching outputs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float* A, const float* B, float* C, int* bad, int tiles) {
  const int lane = threadIdx.x;
  int nbad = 0;
  for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
    const float* a = A + (size_t)t * 32 * 4;   // [row][k]
    const float* b = B + (size_t)t * 32 * 4;   // [col][k]
    const int r = lane & 31, kk = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r * 4 + kk], b[r * 4 + kk], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r * 4 + 2 + kk], b[r * 4 + 2 + kk], acc, 0, 0, 0);
    const int col = lane & 31;
    for (int v = 0; v < 16; ++v) {
      const int row = (v >> 2) * 8 + (lane >> 5) * 4 + (v & 3);
      const float* x = a + row * 4;
      const float* y = b + col * 4;
      float want = __fmaf_rn(x[3], y[3], __fmaf_rn(x[2], y[2], __fmaf_rn(x[1], y[1], __fmul_rn(x[0], y[0]))));
      if (__float_as_uint(want) != __float_as_uint(acc[v])) ++nbad;
      C[((size_t)t * 32 + row) * 32 + col] = acc[v];
    }
  }
  atomicAdd(bad, nbad);
}
int main() {
  const int tiles = 4096;
  size_t n = (size_t)tiles * 128;
  float *hA = (float*)malloc(n * 4), *hB = (float*)malloc(n * 4);
  srand(1);
  for (size_t i = 0; i < n; ++i) {
    hA[i] = (float)rand() / RAND_MAX - 0.5f + ((i & 3) == 2 ? 8.f * (i % 7 == 0) : 0.f);
    hB[i] = (float)rand() / RAND_MAX - 0.5f;
    if ((i & 3) == 3 && (i >> 2) % 2 == 0) hA[i] = hB[i] = 0.f;   // half of the rows / columns: K = 3, the fourth k is zero padding
  }
  float *A, *B, *C; int* bad; int hbad = -1;
  hipMalloc(&A, n * 4); hipMalloc(&B, n * 4); hipMalloc(&C, (size_t)tiles * 1024 * 4); hipMalloc(&bad, 4);
  hipMemcpy(A, hA, n * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB, n * 4, hipMemcpyHostToDevice); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(k, dim3(256), dim3(64), 0, 0, A, B, C, bad, tiles);
  hipMemcpy(&hbad, bad, 4, hipMemcpyDeviceToHost);
  printf("mfma_f32_32x32x2 as a k-ordered fma chain: %d of %zu outputs differ\n", hbad, (size_t)tiles * 1024);
  return 0;
}
