// rccl_comm.hip -- the library's own RCCL communicator (one process per GPU, xGMI).
//
// The reference has no multi-GPU code; BASELINE.json's north_star asks for "RCCL all-reduce of the 6x6 system over xGMI".  Every
// inter-rank exchange of the hot path is an in-place all-reduce of 64-bit integers on a device buffer (exact: owners contribute bit
// patterns / partial fixed-point sums, everybody else zeros) or a broadcast of a frame:
//   op 0  SUM of int64   -- the grouped normal-equation accumulators of a split model after every {ICP || residual} launch
//                           (launch_gn_track's hook), the per-superpixel segmentation sums with the tracked poses in their tail
//   op 1  MIN of uint64  -- the z-keys of a surfel-range sharded index map
// With cf_rccl_init the context owns an ncclComm_t and runs these with ncclAllReduce / ncclBroadcast on the stream the work is
// enqueued on: in place, no staging copies, no callback into the host language.  cf_set_collective (a caller-supplied function)
// stays as the path for process groups that are not RCCL (the gloo tests on a one-GPU box).
#include <rccl/rccl.h>
#include <string.h>

#include "cf_host.h"

namespace {

struct RcclState {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

int fail(cf_ctx* ctx, const char* what, ncclResult_t r)
{
    ctx->set_error(std::string(what) + ": " + ncclGetErrorString(r));
    return CF_EHIP;
}

// cf_ctx::collective signature
int rccl_collective(void* user, int op, void* dev_buf, uint64_t words, void* stream)
{
    cf_ctx* ctx = static_cast<cf_ctx*>(user);
    RcclState* st = static_cast<RcclState*>(ctx->rccl);
    if (!st || !st->comm) return -1;
    const ncclResult_t r = (op == 0) ? ncclAllReduce(dev_buf, dev_buf, words, ncclInt64, ncclSum, st->comm, static_cast<hipStream_t>(stream))
                                     : ncclAllReduce(dev_buf, dev_buf, words, ncclUint64, ncclMin, st->comm, static_cast<hipStream_t>(stream));
    if (r != ncclSuccess) { fail(ctx, "ncclAllReduce", r); return -1; }
    return 0;
}

}  // namespace

extern "C" {

int cf_rccl_unique_id(void* id128)
{
    static_assert(sizeof(ncclUniqueId) == CF_RCCL_ID_BYTES, "ncclUniqueId is 128 bytes");
    if (!id128) return CF_EINVAL;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return CF_EHIP;
    memcpy(id128, &id, sizeof(id));
    return CF_OK;
}

int cf_rccl_init(cf_ctx* ctx, const void* id128, int rank, int world)
{
    if (!ctx || !id128 || world < 1 || rank < 0 || rank >= world) return CF_EINVAL;
    if (ctx->rccl) { ctx->set_error("cf_rccl_init: the context already has a communicator"); return CF_ESTATE; }
    if (hipSetDevice(ctx->cfg.device) != hipSuccess) { ctx->set_error("cf_rccl_init: hipSetDevice failed"); return CF_EHIP; }
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    RcclState* st = new RcclState();
    st->rank = rank; st->world = world;
    const ncclResult_t r = ncclCommInitRank(&st->comm, world, id, rank);
    if (r != ncclSuccess) { delete st; return fail(ctx, "ncclCommInitRank", r); }
    ctx->rccl = st;
    ctx->collective = rccl_collective; ctx->collective_user = ctx;  // the split reductions of the Gauss-Newton loop / index map
    return CF_OK;
}

int cf_rccl_allreduce(cf_ctx* ctx, void* dev_buf, uint64_t words, int op, void* hip_stream)
{
    if (!ctx || !dev_buf || (op != 0 && op != 1)) return CF_EINVAL;
    if (!ctx->rccl) { ctx->set_error("cf_rccl_allreduce: no communicator (cf_rccl_init)"); return CF_ESTATE; }
    return rccl_collective(ctx, op, dev_buf, words, hip_stream ? hip_stream : (void*)ctx->cur()) == 0 ? CF_OK : CF_EHIP;
}

int cf_rccl_broadcast(cf_ctx* ctx, void* dev_buf, uint64_t bytes, int root, void* hip_stream)
{
    if (!ctx || !dev_buf) return CF_EINVAL;
    RcclState* st = static_cast<RcclState*>(ctx->rccl);
    if (!st) { ctx->set_error("cf_rccl_broadcast: no communicator (cf_rccl_init)"); return CF_ESTATE; }
    if (root < 0 || root >= st->world) return CF_EINVAL;
    const ncclResult_t r = ncclBroadcast(dev_buf, dev_buf, bytes, ncclUint8, root, st->comm,
                                         hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->cur());
    if (r != ncclSuccess) return fail(ctx, "ncclBroadcast", r);
    return CF_OK;
}

int cf_rccl_info(const cf_ctx* ctx, int* rank, int* world, int* version)
{
    if (!ctx) return CF_EINVAL;
    const RcclState* st = static_cast<const RcclState*>(ctx->rccl);
    if (rank) *rank = st ? st->rank : -1;
    if (world) *world = st ? st->world : 0;
    if (version) { int v = 0; if (ncclGetVersion(&v) != ncclSuccess) v = 0; *version = v; }
    return st ? CF_OK : CF_ESTATE;
}

int cf_rccl_destroy(cf_ctx* ctx)
{
    if (!ctx) return CF_EINVAL;
    RcclState* st = static_cast<RcclState*>(ctx->rccl);
    if (!st) return CF_OK;
    (void)hipStreamSynchronize(ctx->stream);
    if (st->comm) (void)ncclCommDestroy(st->comm);
    if (ctx->collective == rccl_collective) { ctx->collective = nullptr; ctx->collective_user = nullptr; }
    delete st;
    ctx->rccl = nullptr;
    return CF_OK;
}

}  // extern "C"
