// staging.hip -- restart / diagnostic staging: the prognostic fields leave the GPU for the host arrays
// MOM6 registered for them (register_restart_field, src/framework/MOM_restart.F90; the arrays post_data reads,
// src/framework/MOM_diag_mediator.F90) without stopping the model.  A staged field is first copied device-to-device on the
// compute stream (a snapshot: ~0.3 ms for a 1 GB field, after which the model may overwrite the field), then moved to the
// host on a stream of its own while the next steps run; mom6hip_stage_wait is the only point that blocks, and the host
// side calls it where it needs the data (before save_restart writes, before a diagnostic is averaged).
#include "common.hpp"

namespace m6 {
void staging_destroy(mom6hip_ctx *ctx) {
  if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
  for (auto &s : ctx->stage_slots) {
    s.buf.release();
    if (s.snap) (void)hipEventDestroy(s.snap);
    if (s.done) (void)hipEventDestroy(s.done);
  }
  ctx->stage_slots.clear();
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  ctx->copy_stream = nullptr;
}
}  // namespace m6

extern "C" {

int mom6hip_host_register(void *hptr, uint64_t bytes) {
  M6_REQUIRE(hptr != nullptr && bytes > 0, "mom6hip_host_register: null pointer or no bytes");
  M6_HIP(hipHostRegister(hptr, bytes, hipHostRegisterDefault));
  return 0;
}

int mom6hip_host_unregister(void *hptr) {
  M6_REQUIRE(hptr != nullptr, "mom6hip_host_unregister: null pointer");
  M6_HIP(hipHostUnregister(hptr));
  return 0;
}

int mom6hip_stage_to_host(mom6hip_ctx_t *ctx, void *hptr, const void *dptr, uint64_t bytes) {
  M6_REQUIRE(ctx && hptr && dptr && bytes > 0, "mom6hip_stage_to_host: null argument or no bytes");
  if (!ctx->copy_stream) M6_HIP(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  if (ctx->stage_next == (int)ctx->stage_slots.size()) {
    ctx->stage_slots.emplace_back();
    auto &n = ctx->stage_slots.back();
    M6_HIP(hipEventCreateWithFlags(&n.snap, hipEventDisableTiming));
    M6_HIP(hipEventCreateWithFlags(&n.done, hipEventDisableTiming));
  }
  auto &s = ctx->stage_slots[ctx->stage_next++];
  M6_REQUIRE(s.buf.reserve(bytes) == 0, "mom6hip_stage_to_host: out of device memory for the snapshot");
  // (a slot is reused only after mom6hip_stage_wait, which has seen its last copy land: nothing to order against here)
  M6_HIP(hipMemcpyAsync(s.buf.p, dptr, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  M6_HIP(hipEventRecord(s.snap, ctx->stream));
  M6_HIP(hipStreamWaitEvent(ctx->copy_stream, s.snap, 0));
  M6_HIP(hipMemcpyAsync(hptr, s.buf.p, bytes, hipMemcpyDeviceToHost, ctx->copy_stream));
  M6_HIP(hipEventRecord(s.done, ctx->copy_stream));
  ctx->xfer[2]++; ctx->xfer[3] += bytes;
  return 0;
}

int mom6hip_stage_query(mom6hip_ctx_t *ctx, int32_t *pending) {
  M6_REQUIRE(ctx && pending, "mom6hip_stage_query: null argument");
  int n = 0;
  for (int q = 0; q < ctx->stage_next; q++) {
    const hipError_t e = hipEventQuery(ctx->stage_slots[q].done);
    if (e == hipErrorNotReady) n++;
    else if (e != hipSuccess) { m6::set_error("mom6hip_stage_query: %s", hipGetErrorString(e)); return 1; }
  }
  *pending = n;
  return 0;
}

int mom6hip_stage_wait(mom6hip_ctx_t *ctx) {
  M6_REQUIRE(ctx != nullptr, "mom6hip_stage_wait: null context");
  if (ctx->copy_stream) M6_HIP(hipStreamSynchronize(ctx->copy_stream));
  ctx->stage_next = 0;
  return 0;
}

// ---- the HBM bandwidth this device delivers to plain streaming kernels (what the roofline fractions can be read against) ----
namespace {
__global__ __launch_bounds__(256) void stream_copy_kernel(double2 *__restrict__ a, const double2 *__restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i];
}
__global__ __launch_bounds__(256) void stream_triad_kernel(double2 *__restrict__ a, const double2 *__restrict__ b,
                                                           const double2 *__restrict__ c, double s, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 x = b[i], y = c[i];
    a[i] = make_double2(x.x + s * y.x, x.y + s * y.y);
  }
}
}  // namespace

int mom6hip_stream_bandwidth(mom6hip_ctx_t *ctx, uint64_t bytes_per_array, int32_t reps, double *copy_GBs, double *triad_GBs) {
  M6_REQUIRE(ctx && bytes_per_array >= 4096 && reps >= 1 && copy_GBs && triad_GBs, "mom6hip_stream_bandwidth: bad argument");
  const size_t n = (size_t)(bytes_per_array / sizeof(double2));
  double2 *a = nullptr, *b = nullptr, *c = nullptr;
  M6_HIP(hipMalloc((void **)&a, n * sizeof(double2)));
  if (hipMalloc((void **)&b, n * sizeof(double2)) != hipSuccess || hipMalloc((void **)&c, n * sizeof(double2)) != hipSuccess) {
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
    m6::set_error("mom6hip_stream_bandwidth: out of device memory"); return 1;
  }
  hipStream_t s = ctx->stream;
  (void)hipMemsetAsync(b, 0, n * sizeof(double2), s); (void)hipMemsetAsync(c, 0, n * sizeof(double2), s);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const dim3 grid(256 * 128), block(256);      // 128 blocks a CU, grid-stride (tools/stream_probe.hip: the fastest of the forms tried)
  auto timed = [&](bool triad) -> double {
    if (triad) hipLaunchKernelGGL(stream_triad_kernel, grid, block, 0, s, a, b, c, 3.0, n);      // warm-up
    else hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, s, a, b, n);
    (void)hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) {
      if (triad) hipLaunchKernelGGL(stream_triad_kernel, grid, block, 0, s, a, b, c, 3.0, n);
      else hipLaunchKernelGGL(stream_copy_kernel, grid, block, 0, s, a, b, n);
    }
    (void)hipEventRecord(e1, s);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return (double)(triad ? 3 : 2) * (double)(n * sizeof(double2)) * reps / ((double)ms * 1.0e6);
  };
  *copy_GBs = timed(false);
  *triad_GBs = timed(true);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b); (void)hipFree(c);
  M6_HIP(hipGetLastError());
  return 0;
}

// ---- the cost of a kernel node in a replayed hipGraph: the floor of the barotropic subcycle (bench.py) ----
namespace {
__global__ __launch_bounds__(256) void floor_touch_kernel(double *__restrict__ a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] += 1.0;
}
}  // namespace

int mom6hip_graph_node_floor(mom6hip_ctx_t *ctx, int32_t nodes, int64_t points, int32_t reps, double *us_per_node) {
  M6_REQUIRE(ctx && nodes >= 1 && nodes <= 4096 && points >= 1 && reps >= 1 && us_per_node, "mom6hip_graph_node_floor: bad argument");
  double *a = nullptr;
  M6_HIP(hipMalloc((void **)&a, (size_t)points * sizeof(double)));
  hipStream_t s = ctx->stream;
  (void)hipMemsetAsync(a, 0, (size_t)points * sizeof(double), s);
  hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
  const dim3 grid((unsigned)((points + 255) / 256)), block(256);
  hipStream_t cap = nullptr;      // (the context's stream may be the null stream, which cannot be captured)
  M6_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
  M6_HIP(hipStreamBeginCapture(cap, hipStreamCaptureModeRelaxed));
  for (int n = 0; n < nodes; n++) hipLaunchKernelGGL(floor_touch_kernel, grid, block, 0, cap, a, (size_t)points);
  M6_HIP(hipStreamEndCapture(cap, &graph));
  (void)hipStreamDestroy(cap);
  M6_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  M6_HIP(hipGraphLaunch(exec, s));      // warm-up
  (void)hipEventRecord(e0, s);
  for (int r = 0; r < reps; r++) (void)hipGraphLaunch(exec, s);
  (void)hipEventRecord(e1, s);
  (void)hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_node = (double)ms * 1.0e3 / ((double)reps * nodes);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipGraphExecDestroy(exec); (void)hipGraphDestroy(graph);
  (void)hipFree(a);
  M6_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
