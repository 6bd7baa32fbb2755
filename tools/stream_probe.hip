// tools/stream_probe.hip -- how fast plain streaming kernels run on the box (variants of the copy / triad of
// mom6hip_stream_bandwidth: vector width, unroll, grid size), to choose the form bench.py reports as "measured".
// hipcc --offload-arch=gfx950 -O3 tools/stream_probe.hip -o gpurun_out/stream_probe && gpurun_out/stream_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int U> __global__ __launch_bounds__(256) void copy_k(double2 *__restrict__ a, const double2 *__restrict__ b, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    double2 x[U];
#pragma unroll
    for (int q = 0; q < U; q++) x[q] = b[i + q * stride];
#pragma unroll
    for (int q = 0; q < U; q++) a[i + q * stride] = x[q];
  }
  for (; i < n; i += stride) a[i] = b[i];
}
template <int U> __global__ __launch_bounds__(256) void triad_k(double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c, double s, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    double2 x[U], y[U];
#pragma unroll
    for (int q = 0; q < U; q++) { x[q] = b[i + q * stride]; y[q] = c[i + q * stride]; }
#pragma unroll
    for (int q = 0; q < U; q++) a[i + q * stride] = make_double2(x[q].x + s * y[q].x, x[q].y + s * y[q].y);
  }
  for (; i < n; i += stride) a[i] = make_double2(b[i].x + s * c[i].x, b[i].y + s * c[i].y);
}
template <class F> double timeit(F f, double bytes) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipEventRecord(e0); for (int r = 0; r < 10; r++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return bytes * 10 / (ms * 1e6);
}
int main() {
  for (size_t gib : {1, 4}) {
    const size_t n = (gib << 30) / 16;
    double2 *a, *b, *c; hipMalloc(&a, n * 16); hipMalloc(&b, n * 16); hipMalloc(&c, n * 16);
    hipMemset(b, 0, n * 16); hipMemset(c, 0, n * 16);
    for (int blocks : {256 * 8, 256 * 32, 256 * 128}) {
      printf("%zu GiB arrays, %d blocks: copy U1 %.0f U4 %.0f  triad U1 %.0f U4 %.0f  hipMemcpyDtoD %.0f GB/s\n", gib, blocks,
             timeit([&] { copy_k<1><<<blocks, 256>>>(a, b, n); }, 2.0 * n * 16), timeit([&] { copy_k<4><<<blocks, 256>>>(a, b, n); }, 2.0 * n * 16),
             timeit([&] { triad_k<1><<<blocks, 256>>>(a, b, c, 3.0, n); }, 3.0 * n * 16),
             timeit([&] { triad_k<4><<<blocks, 256>>>(a, b, c, 3.0, n); }, 3.0 * n * 16),
             timeit([&] { hipMemcpyAsync(a, b, n * 16, hipMemcpyDeviceToDevice, 0); }, 2.0 * n * 16));
    }
    hipFree(a); hipFree(b); hipFree(c);
  }
  return 0;
}
