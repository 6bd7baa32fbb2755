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

}  // extern "C"
