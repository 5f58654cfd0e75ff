// Stream-order probe: kernel A spins, then writes 1 into buf; kernel B (same stream, launched right behind it) counts the
// entries that are still 0.  In-order execution => the count is always 0.
#include <hip/hip_runtime.h>
__global__ void slow_writer(int* buf, int n, long spin) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  long t0 = clock64();
  while (clock64() - t0 < spin) {}
  if (i < n) buf[i] = 1;
}
__global__ void reader(const int* buf, int n, int* zeros) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && buf[i] == 0) atomicAdd(zeros, 1);
}
__global__ void clear(int* buf, int n, int* zeros) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = 0;
  if (i == 0) *zeros = 0;
}
extern "C" int order_probe(int* buf, int n, int* zeros, long spin, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(clear, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, zeros);
  hipLaunchKernelGGL(slow_writer, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, spin);
  hipLaunchKernelGGL(reader, dim3((n + 255) / 256), dim3(256), 0, s, buf, n, zeros);
  return (int)hipGetLastError();
}
