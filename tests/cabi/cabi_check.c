/* The drop-in boundary exercised from plain C: no Python, no torch -- hipMalloc'd buffers, the entry points of
 * include/sam6d_hip.h, and the C oracle (oracle/pointops_oracle.c, test infrastructure) as the checker.
 * Built by tests/cabi/Makefile (gcc), run on a GPU box by tests/test_cabi_c_gpu.py.  Exit code 0 = every check passed. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/sam6d_hip.h"

/* oracle/pointops_oracle.c (EXT/src/sampling.cpp:76-118, ball_query.cpp:16-62, group_points.cpp:20-45) */
void orc_furthest_point_sampling(int b, int n, int m, const float* dataset, int* idxs);
void orc_ball_query(int b, int n, int m, float radius, int nsample, const float* new_xyz, const float* xyz, int* idx);
void orc_group_points(int b, int c, int n, int npoints, int nsample, const float* points, const int* idx, float* out);

#define HIP_OK(e)                                                                          \
  do {                                                                                     \
    hipError_t _e = (e);                                                                   \
    if (_e != hipSuccess) {                                                                \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(_e), __FILE__, __LINE__); \
      return 2;                                                                            \
    }                                                                                      \
  } while (0)
#define SAM_OK(e)                                                                      \
  do {                                                                                 \
    int _r = (e);                                                                      \
    if (_r != 0) {                                                                     \
      fprintf(stderr, "%s failed (rc=%d): %s\n", #e, _r, sam6d_last_error());          \
      return 3;                                                                        \
    }                                                                                  \
  } while (0)

static unsigned int lcg_state = 12345u;
static float frand(void) {  /* uniform in [-0.5, 0.5) */
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)(lcg_state >> 8) / 16777216.0f - 0.5f;
}

static int compare_i32(const char* what, const int* got, const int* want, size_t n) {
  for (size_t i = 0; i < n; ++i)
    if (got[i] != want[i]) {
      fprintf(stderr, "%s: mismatch at %zu: got %d want %d\n", what, i, got[i], want[i]);
      return 1;
    }
  printf("  %-44s %zu values bit-exact\n", what, n);
  return 0;
}

int main(void) {
  int ndev = 0;
  HIP_OK(hipGetDeviceCount(&ndev));
  if (ndev < 1) {
    fprintf(stderr, "no HIP device\n");
    return 2;
  }
  HIP_OK(hipSetDevice(0));
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));
  printf("cabi_check: ABI version %d\n", sam6d_abi_version());
  if (sam6d_abi_version() != SAM6D_ABI_VERSION) {
    fprintf(stderr, "library ABI %d != header ABI %d\n", sam6d_abi_version(), SAM6D_ABI_VERSION);
    return 3;
  }

  /* ---- clouds: B = 2, N = 5000 (> 4096: the multi-workgroup FPS with its scratch row), 256 samples */
  const int B = 2, N = 5000, M = 256, NS = 32, C = 5;
  const float radius = 0.12f;
  float* xyz = (float*)malloc(sizeof(float) * B * N * 3);
  for (int i = 0; i < B * N * 3; ++i) xyz[i] = frand();
  for (int i = 0; i < N; i += 9) { xyz[i * 3] *= 0.02f; xyz[i * 3 + 1] *= 0.02f; xyz[i * 3 + 2] *= 0.02f; } /* origin-ball skips */
  float *d_xyz, *d_temp, *d_q, *d_feat, *d_grouped;
  int *d_fps, *d_bq;
  HIP_OK(hipMalloc((void**)&d_xyz, sizeof(float) * B * N * 3));
  HIP_OK(hipMalloc((void**)&d_temp, sizeof(float) * B * N));
  HIP_OK(hipMalloc((void**)&d_fps, sizeof(int) * B * M));
  HIP_OK(hipMemcpyAsync(d_xyz, xyz, sizeof(float) * B * N * 3, hipMemcpyHostToDevice, stream));
  SAM_OK(sam6d_furthest_point_sampling(d_xyz, B, N, M, d_temp, d_fps, stream));
  int* fps = (int*)malloc(sizeof(int) * B * M);
  HIP_OK(hipMemcpyAsync(fps, d_fps, sizeof(int) * B * M, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int* fps_want = (int*)calloc((size_t)B * M, sizeof(int));
  orc_furthest_point_sampling(B, N, M, xyz, fps_want);
  int bad = compare_i32("sam6d_furthest_point_sampling (N=5000)", fps, fps_want, (size_t)B * M);

  /* ---- ball query around the sampled points, then grouping of a (B,C,N) feature map */
  float* q = (float*)malloc(sizeof(float) * B * M * 3);
  for (int b = 0; b < B; ++b)
    for (int j = 0; j < M; ++j)
      for (int c = 0; c < 3; ++c) q[((size_t)b * M + j) * 3 + c] = xyz[((size_t)b * N + fps_want[b * M + j]) * 3 + c];
  HIP_OK(hipMalloc((void**)&d_q, sizeof(float) * B * M * 3));
  HIP_OK(hipMalloc((void**)&d_bq, sizeof(int) * B * M * NS));
  HIP_OK(hipMemcpyAsync(d_q, q, sizeof(float) * B * M * 3, hipMemcpyHostToDevice, stream));
  SAM_OK(sam6d_ball_query(d_q, d_xyz, B, N, M, radius, NS, d_bq, stream));
  int* bq = (int*)malloc(sizeof(int) * B * M * NS);
  HIP_OK(hipMemcpyAsync(bq, d_bq, sizeof(int) * B * M * NS, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  int* bq_want = (int*)calloc((size_t)B * M * NS, sizeof(int));
  orc_ball_query(B, N, M, radius, NS, q, xyz, bq_want);
  bad |= compare_i32("sam6d_ball_query (r=0.12, 32 samples)", bq, bq_want, (size_t)B * M * NS);

  float* feat = (float*)malloc(sizeof(float) * B * C * N);
  for (int i = 0; i < B * C * N; ++i) feat[i] = frand();
  HIP_OK(hipMalloc((void**)&d_feat, sizeof(float) * B * C * N));
  HIP_OK(hipMalloc((void**)&d_grouped, sizeof(float) * B * C * M * NS));
  HIP_OK(hipMemcpyAsync(d_feat, feat, sizeof(float) * B * C * N, hipMemcpyHostToDevice, stream));
  SAM_OK(sam6d_group_points(d_feat, d_bq, B, C, N, M, NS, d_grouped, stream));
  float* grouped = (float*)malloc(sizeof(float) * B * C * M * NS);
  HIP_OK(hipMemcpyAsync(grouped, d_grouped, sizeof(float) * B * C * M * NS, hipMemcpyDeviceToHost, stream));
  HIP_OK(hipStreamSynchronize(stream));
  float* grouped_want = (float*)calloc((size_t)B * C * M * NS, sizeof(float));
  orc_group_points(B, C, N, M, NS, feat, bq_want, grouped_want);
  if (memcmp(grouped, grouped_want, sizeof(float) * B * C * M * NS) != 0) {
    fprintf(stderr, "sam6d_group_points: bytes differ\n");
    bad = 1;
  } else {
    printf("  %-44s %d values bit-exact\n", "sam6d_group_points", B * C * M * NS);
  }

  /* ---- error contract: non-zero return + a message, nothing launched */
  const int rc = sam6d_ball_query(NULL, d_xyz, B, N, M, radius, NS, d_bq, stream);
  if (rc == 0 || strlen(sam6d_last_error()) == 0) {
    fprintf(stderr, "null pointer was not rejected\n");
    bad = 1;
  } else {
    printf("  null pointer rejected: rc=%d \"%s\"\n", rc, sam6d_last_error());
  }

  hipFree(d_xyz); hipFree(d_temp); hipFree(d_fps); hipFree(d_q); hipFree(d_bq); hipFree(d_feat); hipFree(d_grouped);
  hipStreamDestroy(stream);
  free(xyz); free(fps); free(fps_want); free(q); free(bq); free(bq_want); free(feat); free(grouped); free(grouped_want);
  if (bad) return 1;
  printf("cabi_check: ok\n");
  return 0;
}
