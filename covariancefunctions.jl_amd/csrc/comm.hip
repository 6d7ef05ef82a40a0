// comm.hip — the ONE collective of the multi-GPU MVM behind the C ABI (round 5): a ctx can own an RCCL communicator, and
//   covgram_mvm_sharded        = this rank's row shard of G a (the single-GPU kernels) + ncclAllGather of b
//   covgram_mvm_sym_allreduce  = this rank's cyclic panels of the upper triangle (covgram_mvm_sym_partial) + ncclAllReduce
// both ENQUEUED ON THE CTX STREAM right behind the kernels — no second stream, no event hops, no host synchronisation — which is the
// GPU analogue of the reference's one parallel axis, `@threads for i in 1:n` over output rows (src/gramian.jl:78-87): rows of G shard
// naturally, the weights a and all column points stay replicated, and the gathered b is the replicated a of the next
// Krylov iteration.  One process per GPU; the caller moves the 128-byte unique id between its processes however it likes (MPI.jl,
// torch.distributed's store, a file) — the only thing that ever crosses processes outside RCCL.
//
// RCCL is resolved at first use (dlopen of librccl.so.1: in a PyTorch process that is the copy torch already loaded), so the library has
// no link-time dependency on it and single-GPU callers never touch it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include "common.hpp"

namespace covgram {

struct RcclApi {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

static RcclApi* rccl() {
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
            if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather || !api.AllReduce) { dlclose(api.handle); api.handle = nullptr; }
        }
    }
    return api.handle ? &api : nullptr;
}

#define CG_CHECK_RCCL(api, expr)                                                                                            \
    do {                                                                                                                    \
        ncclResult_t _r = (expr);                                                                                           \
        if (_r != ncclSuccess) {                                                                                            \
            set_error("%s failed: %s", #expr, (api)->GetErrorString ? (api)->GetErrorString(_r) : "RCCL error");            \
            return COVGRAM_EHIP;                                                                                            \
        }                                                                                                                   \
    } while (0)

static ncclDataType_t nccl_type(int dtype) { return dtype == COVGRAM_F64 ? ncclDouble : ncclFloat; }

int comm_destroy(covgram_ctx* ctx) {
    if (ctx->comm) {
        RcclApi* api = rccl();
        if (api) (void)api->CommDestroy((ncclComm_t)ctx->comm);
        ctx->comm = nullptr;
    }
    ctx->comm_rank = 0; ctx->comm_world = 0;
    return COVGRAM_OK;
}

// y[i] <- alpha * t[i] + beta * y[i]  (beta == 0: y's old contents, NaN included, are ignored — src/gramian.jl:80)
template <typename T>
__global__ __launch_bounds__(256) void axpby_kernel(T* __restrict__ y, const T* __restrict__ t, int64_t n, T alpha, T beta) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    T v = alpha * t[i];
    if (beta != (T)0) v = fma_t(beta, y[i], v);
    y[i] = v;
}

}  // namespace covgram

using namespace covgram;

extern "C" {

int covgram_comm_unique_id(void* id, int64_t bytes) {
    CG_REQUIRE(id != nullptr && bytes >= (int64_t)NCCL_UNIQUE_ID_BYTES, COVGRAM_EINVAL, "unique id buffer must hold %d bytes", (int)NCCL_UNIQUE_ID_BYTES);
    RcclApi* api = rccl();
    CG_REQUIRE(api != nullptr, COVGRAM_EUNSUPPORTED, "librccl.so.1 could not be loaded: %s", dlerror() ? dlerror() : "not found");
    ncclUniqueId uid;
    CG_CHECK_RCCL(api, api->GetUniqueId(&uid));
    memcpy(id, &uid, sizeof(uid));
    return COVGRAM_OK;
}

int covgram_comm_create(covgram_ctx* ctx, const void* unique_id, int32_t rank, int32_t world) {
    CG_REQUIRE(ctx && unique_id, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(world >= 1 && rank >= 0 && rank < world, COVGRAM_EINVAL, "rank %d of %d", rank, world);
    CG_REQUIRE(ctx->comm == nullptr, COVGRAM_EINVAL, "this ctx already owns a communicator (covgram_comm_destroy first)");
    RcclApi* api = rccl();
    CG_REQUIRE(api != nullptr, COVGRAM_EUNSUPPORTED, "librccl.so.1 could not be loaded");
    CG_DEVICE(ctx);
    ncclUniqueId uid;
    memcpy(&uid, unique_id, sizeof(uid));
    ncclComm_t comm = nullptr;
    CG_CHECK_RCCL(api, api->CommInitRank(&comm, world, uid, rank));      // collective: every rank of the job calls it
    ctx->comm = (void*)comm; ctx->comm_rank = rank; ctx->comm_world = world;
    return COVGRAM_OK;
}

int covgram_comm_destroy(covgram_ctx* ctx) {
    CG_REQUIRE(ctx, COVGRAM_EINVAL, "NULL argument");
    if (ctx->comm) { CG_DEVICE(ctx); CG_CHECK_HIP(hipStreamSynchronize(ctx->stream)); }
    return comm_destroy(ctx);
}

int covgram_comm_info(const covgram_ctx* ctx, int32_t* rank, int32_t* world) {
    CG_REQUIRE(ctx, COVGRAM_EINVAL, "NULL argument");
    if (rank) *rank = ctx->comm ? ctx->comm_rank : 0;
    if (world) *world = ctx->comm ? ctx->comm_world : 0;               // 0: no communicator
    return COVGRAM_OK;
}

// recv[r * count .. (r + 1) * count) <- send of rank r, on the ctx stream; in place when send == recv + rank * count
int covgram_comm_all_gather(covgram_ctx* ctx, const void* send, void* recv, int64_t count, int32_t dtype) {
    CG_REQUIRE(ctx && ctx->comm, COVGRAM_EINVAL, "this ctx owns no communicator (covgram_comm_create)");
    CG_REQUIRE(count >= 0 && (count == 0 || (send && recv)), COVGRAM_EINVAL, "bad all-gather arguments");
    if (count == 0) return COVGRAM_OK;
    RcclApi* api = rccl();
    CG_DEVICE(ctx);
    CG_CHECK_RCCL(api, api->AllGather(send, recv, (size_t)count, nccl_type(dtype), (ncclComm_t)ctx->comm, ctx->stream));
    return COVGRAM_OK;
}

int covgram_comm_all_reduce_sum(covgram_ctx* ctx, void* buf, int64_t count, int32_t dtype) {
    CG_REQUIRE(ctx && ctx->comm, COVGRAM_EINVAL, "this ctx owns no communicator (covgram_comm_create)");
    CG_REQUIRE(count >= 0 && (count == 0 || buf), COVGRAM_EINVAL, "bad all-reduce arguments");
    if (count == 0) return COVGRAM_OK;
    RcclApi* api = rccl();
    CG_DEVICE(ctx);
    CG_CHECK_RCCL(api, api->AllReduce(buf, buf, (size_t)count, nccl_type(dtype), ncclSum, (ncclComm_t)ctx->comm, ctx->stream));
    return COVGRAM_OK;
}

// y (n entries, complete on every rank) <- alpha * G(k; X, Y) * a + beta * y.  X: ALL n row points, Y: all m column points, a: the m
// weights — replicated on every rank, device memory.  Rank r evaluates the rows [r per, (r + 1) per), per = ceil(n / world); when world
// divides n it writes them straight into its slice of y and the all-gather runs in place, otherwise through two workspace buffers.
int covgram_mvm_sharded(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const covgram_points* Y, const void* a, void* y,
                        double alpha, double beta) {
    CG_REQUIRE(ctx && X && Y, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(ctx->comm != nullptr, COVGRAM_EINVAL, "this ctx owns no communicator (covgram_comm_create)");
    const int world = ctx->comm_world, rank = ctx->comm_rank;
    const int64_t n = X->n, per = (n + world - 1) / world;
    const int64_t lo = std::min<int64_t>(n, (int64_t)rank * per), hi = std::min<int64_t>(n, lo + per);
    const size_t ts = dtype_size(X->dtype);
    if (n == 0) return COVGRAM_OK;
    CG_REQUIRE(y != nullptr, COVGRAM_EINVAL, "y is NULL");
    covgram_points* Xs = nullptr;
    int rc = COVGRAM_OK;
    if (hi > lo) { rc = covgram_points_slice(X, lo, hi - lo, &Xs); if (rc) return rc; }
    const bool exact = per * world == n;
    if (exact) {
        rc = covgram_mvm(ctx, k, Xs, Y, a, Y->n, (char*)y + (size_t)lo * ts, per, 1, alpha, beta, COVGRAM_DEVICE);
        if (!rc) rc = covgram_comm_all_gather(ctx, (const char*)y + (size_t)lo * ts, y, per, X->dtype);
    } else {
        CG_DEVICE(ctx);
        void* buf;                                                           // [per: this rank's shard, zero padded][per * world: gathered]
        rc = ws_reserve(ctx, 2, (size_t)per * (size_t)(world + 1) * ts, &buf);
        if (!rc) {
            char* shard = (char*)buf; char* full = shard + (size_t)per * ts;
            if (hi - lo < per && hipMemsetAsync(shard, 0, (size_t)per * ts, ctx->stream) != hipSuccess) rc = COVGRAM_EHIP;
            if (!rc && Xs) rc = covgram_mvm(ctx, k, Xs, Y, a, Y->n, shard, per, 1, 1.0, 0.0, COVGRAM_DEVICE);
            if (!rc) rc = covgram_comm_all_gather(ctx, shard, full, per, X->dtype);
            if (!rc) {
                const unsigned grid = (unsigned)((n + 255) / 256);
                if (X->dtype == COVGRAM_F32) hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (float*)y, (const float*)full, n, (float)alpha, (float)beta);
                else hipLaunchKernelGGL(axpby_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (double*)y, (const double*)full, n, alpha, beta);
                if (hipGetLastError() != hipSuccess) { set_error("axpby launch failed"); rc = COVGRAM_EHIP; }
            }
        }
    }
    if (Xs) (void)covgram_points_destroy(Xs);
    return rc;
}

// The same product in the symmetric form of gramian(k, x): this rank's cyclic panels of the upper triangle, then ONE all-reduce.
// COVGRAM_EUNSUPPORTED when no symmetric kernel applies (covgram_mvm_sym_supported): the caller takes covgram_mvm_sharded.
int covgram_mvm_sym_allreduce(covgram_ctx* ctx, const covgram_kernel* k, const covgram_points* X, const void* a, void* y, double alpha, double beta) {
    CG_REQUIRE(ctx && X, COVGRAM_EINVAL, "NULL argument");
    CG_REQUIRE(ctx->comm != nullptr, COVGRAM_EINVAL, "this ctx owns no communicator (covgram_comm_create)");
    const int64_t n = X->n;
    if (n == 0) return COVGRAM_OK;
    const size_t ts = dtype_size(X->dtype);
    int32_t ok = 0;
    int rc = covgram_mvm_sym_supported(ctx, k, X, ctx->comm_world, &ok);
    if (rc) return rc;
    CG_REQUIRE(ok != 0, COVGRAM_EUNSUPPORTED, "no symmetric kernel applies to this kernel / point set / world size");
    if (alpha == 1.0 && beta == 0.0) {
        rc = covgram_mvm_sym_partial(ctx, k, X, a, y, ctx->comm_rank, ctx->comm_world);
        if (!rc) rc = covgram_comm_all_reduce_sum(ctx, y, n, X->dtype);
        return rc;
    }
    CG_DEVICE(ctx);
    void* part;
    rc = ws_reserve(ctx, 2, (size_t)n * ts, &part); if (rc) return rc;
    rc = covgram_mvm_sym_partial(ctx, k, X, a, part, ctx->comm_rank, ctx->comm_world); if (rc) return rc;
    rc = covgram_comm_all_reduce_sum(ctx, part, n, X->dtype); if (rc) return rc;
    const unsigned grid = (unsigned)((n + 255) / 256);
    if (X->dtype == COVGRAM_F32) hipLaunchKernelGGL(axpby_kernel<float>, dim3(grid), dim3(256), 0, ctx->stream, (float*)y, (const float*)part, n, (float)alpha, (float)beta);
    else hipLaunchKernelGGL(axpby_kernel<double>, dim3(grid), dim3(256), 0, ctx->stream, (double*)y, (const double*)part, n, alpha, beta);
    CG_CHECK_HIP(hipGetLastError());
    return COVGRAM_OK;
}

}  // extern "C"
