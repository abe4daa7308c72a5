// comm.hip -- RCCL (over xGMI) time-stack reduction, the only exchange step
// of the path: every rank holds partial stacks [P][L] float32.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <cstring>

#include "../../include/xcx.h"
#include "xcx_internal.h"

using namespace xcx;

static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is expected to be 128 bytes");

#define NCCL_TRY(expr)                                                      \
  do {                                                                      \
    ncclResult_t r_ = (expr);                                               \
    if (r_ != ncclSuccess)                                                  \
      return set_err(XCX_ERCCL, "%s: %s", #expr, ncclGetErrorString(r_));   \
  } while (0)

extern "C" int xcx_comm_unique_id(char* id_bytes128) {
  if (!id_bytes128) return set_err(XCX_EINVAL, "null id buffer");
  ncclUniqueId id;
  NCCL_TRY(ncclGetUniqueId(&id));
  memcpy(id_bytes128, &id, sizeof id);
  return XCX_OK;
}

extern "C" int xcx_comm_init(xcx_ctx* ctx, const char* id_bytes128, int nranks, int rank) {
  if (!ctx || !id_bytes128) return set_err(XCX_EINVAL, "null argument");
  if (nranks < 1 || rank < 0 || rank >= nranks) return set_err(XCX_EINVAL, "bad rank/nranks");
  if (ctx->comm) return set_err(XCX_EINVAL, "communicator already initialised");
  if (hipSetDevice(ctx->device) != hipSuccess) return set_err(XCX_EHIP, "hipSetDevice failed");
  ncclUniqueId id;
  memcpy(&id, id_bytes128, sizeof id);
  ncclComm_t comm;
  NCCL_TRY(ncclCommInitRank(&comm, nranks, id, rank));
  ctx->comm = comm;
  ctx->nranks = nranks;
  ctx->rank = rank;
  return XCX_OK;
}

extern "C" int xcx_comm_destroy(xcx_ctx* ctx) {
  if (!ctx || !ctx->comm) return XCX_OK;
  ncclCommDestroy((ncclComm_t)ctx->comm);
  ctx->comm = nullptr;
  return XCX_OK;
}

extern "C" int xcx_comm_info(xcx_ctx* ctx, int* nranks, int* rank) {
  if (!ctx || !nranks || !rank) return set_err(XCX_EINVAL, "null argument");
  if (!ctx->comm) return set_err(XCX_EINVAL, "xcx_comm_info: xcx_comm_init() not called on this context");
  NCCL_TRY(ncclCommCount((ncclComm_t)ctx->comm, nranks));
  NCCL_TRY(ncclCommUserRank((ncclComm_t)ctx->comm, rank));
  return XCX_OK;
}

namespace xcx {
int comm_reduce_scatter(xcx_ctx* ctx, const float* send, float* recv, size_t per) {
  if (!ctx->comm) return set_err(XCX_EINVAL, "xcx_comm_init() not called on this context");
  NCCL_TRY(ncclReduceScatter(send, recv, per, ncclFloat, ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
  return XCX_OK;
}
}  // namespace xcx

extern "C" int xcx_stack_reduce(xcx_ctx* ctx, float* stack, size_t count, int reduce_scatter) {
  if (!ctx || !stack) return set_err(XCX_EINVAL, "null argument");
  // No communicator is an error even for one rank: a multi-rank job whose xcx_comm_init failed
  // or was forgotten would otherwise get its partial stack back as if it were the total.
  // (A single-rank world calls xcx_comm_init(ctx, id, 1, 0) like any other.)
  if (!ctx->comm) return set_err(XCX_EINVAL, "xcx_stack_reduce: xcx_comm_init() not called on this context");
  ncclComm_t comm = (ncclComm_t)ctx->comm;
  if (reduce_scatter) {
    if (count % ctx->nranks) return set_err(XCX_EINVAL, "count not divisible by nranks");
    const size_t per = count / ctx->nranks;
    NCCL_TRY(ncclReduceScatter(stack, stack + (size_t)ctx->rank * per, per, ncclFloat, ncclSum, comm, ctx->stream));
  } else {
    NCCL_TRY(ncclAllReduce(stack, stack, count, ncclFloat, ncclSum, comm, ctx->stream));
  }
  return XCX_OK;
}
