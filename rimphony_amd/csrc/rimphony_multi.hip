// rimphony_multi.hip -- the multi-GPU half of the C ABI below Python (lib.rs:178-191 is the unit of
// work that shards):
//
//   rimphony_batch_compute_multi_device   one context per device, DEVICE buffers per device, asynchronous launches on every
//                                         device from the calling thread (no host staging, no host threads)
//   rimphony_rccl_*                       the one collective of the path -- the gather of the per-rank shards of the
//                                         interleaved [n][8] table on a root rank -- over RCCL (xGMI between the GPUs of a
//                                         node), for a host that is not Python: librccl is dlopen'ed on first use, the
//                                         library has no link-time dependency on it, and a process that never calls these
//                                         entries never loads it.
//
// Sharding convention (the same as rimphony_batch_compute_multi and rimphony_amd/sharding.py): row i of the table belongs to
// rank i mod world and is row i / world of that rank's shard.
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include "rimphony_internal.h"

// ---- device-buffer multi entry ---------------------------------------------------------------------------------

extern "C" int rimphony_batch_compute_multi_device(rimphony_ctx *const *ctxs, int n_ctx, int dist_kind, const size_t *n_local,
                                                   const double *const *d_s, const double *const *d_theta,
                                                   const double *const *const *d_params, uint32_t coeff_mask, int precision,
                                                   double *const *d_out, int32_t *const *d_status, uint64_t *const *d_work,
                                                   void *const *streams, int synchronize)
{
    if (!ctxs || n_ctx < 1 || n_ctx > 64 || !n_local || !d_s || !d_theta || !d_params || !d_out) return RIMPHONY_EINVAL;
    for (int r = 0; r < n_ctx; r++)
        if (!ctxs[r] || (n_local[r] && (!d_s[r] || !d_theta[r] || !d_params[r] || !d_out[r]))) return RIMPHONY_EINVAL;
    // launches are asynchronous: every device is busy before the first one is waited for
    int launched = 0, launch_rc = RIMPHONY_OK;
    std::string launch_err;
    for (int r = 0; r < n_ctx; r++, launched++) {
        if (n_local[r] == 0) continue;
        launch_rc = rimphony_batch_compute_device_ex(ctxs[r], dist_kind, n_local[r], d_s[r], d_theta[r], d_params[r], coeff_mask,
                                                     precision, d_out[r], d_status ? d_status[r] : nullptr,
                                                     d_work ? d_work[r] : nullptr, streams ? streams[r] : nullptr);
        if (launch_rc) { launch_err = rimphony_last_error(); break; }
    }
    if (launch_rc) {
        // a context refused its launch: the devices before it are already computing into the caller's buffers -- wait for
        // them whatever `synchronize` says, so that "the call failed" also means "nothing is still running"
        for (int r = 0; r < launched; r++) {
            int dev = -1;
            if (n_local[r] == 0 || rimphony_ctx_device(ctxs[r], &dev) != RIMPHONY_OK) continue;
            if (hipSetDevice(dev) == hipSuccess) (void) hipStreamSynchronize(streams ? (hipStream_t) streams[r] : (hipStream_t) 0);
        }
        if (!launch_err.empty()) rim_set_last_error("rimphony_batch_compute_multi_device", launch_err.c_str());
        return launch_rc;
    }
    if (synchronize) {
        int dev0 = 0;
        HIP_TRY(hipGetDevice(&dev0));
        for (int r = 0; r < n_ctx; r++) {
            if (n_local[r] == 0) continue;
            int dev = -1;
            const int rc = rimphony_ctx_device(ctxs[r], &dev);
            if (rc) return rc;
            HIP_TRY(hipSetDevice(dev));
            HIP_TRY(hipStreamSynchronize(streams ? (hipStream_t) streams[r] : (hipStream_t) 0));
        }
        HIP_TRY(hipSetDevice(dev0));
    }
    return RIMPHONY_OK;
}

// ---- RCCL through dlopen -------------------------------------------------------------------------------------------

namespace {

// the few declarations of rccl.h this file needs (ncclResult_t and ncclDataType_t are ints in the ABI; ncclUniqueId is
// 128 bytes; ncclFloat64 = 8)
typedef struct { char internal[128]; } RcclUniqueId;
typedef int (*fn_get_unique_id)(RcclUniqueId *);
typedef int (*fn_comm_init_rank)(void **, int, RcclUniqueId, int);
typedef int (*fn_comm_destroy)(void *);
typedef int (*fn_group)(void);
typedef int (*fn_sendrecv)(const void *, size_t, int, int, void *, hipStream_t);
typedef int (*fn_recv)(void *, size_t, int, int, void *, hipStream_t);
typedef const char *(*fn_errstr)(int);

struct Rccl {
    void *handle = nullptr;
    fn_get_unique_id get_unique_id = nullptr;
    fn_comm_init_rank comm_init_rank = nullptr;
    fn_comm_destroy comm_destroy = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    fn_sendrecv send = nullptr;
    fn_recv recv = nullptr;
    fn_errstr errstr = nullptr;
    std::string why;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void rccl_load()
{
    // RIMPHONY_RCCL_LIB names the library explicitly; otherwise the soname (which also finds a copy that the process --
    // e.g. PyTorch -- has already loaded), then ROCm's default location
    const char *cands[4] = { getenv("RIMPHONY_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1" };
    for (int i = 0; i < 4 && !g_rccl.handle; i++)
        if (cands[i] && *cands[i]) g_rccl.handle = dlopen(cands[i], RTLD_NOW | RTLD_GLOBAL);
    if (!g_rccl.handle) { g_rccl.why = std::string("loading librccl.so.1: ") + (dlerror() ? dlerror() : "not found"); return; }
#define RCCL_SYM(field, type, name) g_rccl.field = (type) dlsym(g_rccl.handle, name); \
    if (!g_rccl.field) { g_rccl.why = std::string("librccl lacks ") + name; g_rccl.handle = nullptr; return; }
    RCCL_SYM(get_unique_id, fn_get_unique_id, "ncclGetUniqueId")
    RCCL_SYM(comm_init_rank, fn_comm_init_rank, "ncclCommInitRank")
    RCCL_SYM(comm_destroy, fn_comm_destroy, "ncclCommDestroy")
    RCCL_SYM(group_start, fn_group, "ncclGroupStart")
    RCCL_SYM(group_end, fn_group, "ncclGroupEnd")
    RCCL_SYM(send, fn_sendrecv, "ncclSend")
    RCCL_SYM(recv, fn_recv, "ncclRecv")
    RCCL_SYM(errstr, fn_errstr, "ncclGetErrorString")
#undef RCCL_SYM
}

// 0 when librccl is usable; RIMPHONY_ENOTSUP (with the reason as the thread's last error) otherwise
int rccl_ready()
{
    std::call_once(g_rccl_once, rccl_load);
    if (g_rccl.handle) return RIMPHONY_OK;
    rim_set_last_error("librccl", g_rccl.why.c_str());
    return RIMPHONY_ENOTSUP;
}

#define RCCL_TRY(expr)                                                           \
    do {                                                                         \
        const int e_ = (expr);                                                   \
        if (e_ != 0) {                                                           \
            rim_set_last_error(#expr, g_rccl.errstr(e_));                        \
            return RIMPHONY_ERCCL;                                               \
        }                                                                        \
    } while (0)

// rows of rank r's shard of an n-row interleaved table
inline size_t shard_rows(size_t n, int r, int world) { return (n + (size_t) world - 1 - (size_t) r) / (size_t) world; }

// blocks[off_r + j][slot] -> table[r + j world][slot]: rank r's contiguous shard back to its rows of the table
__global__ void uninterleave_kernel(const double *blocks, double *table, size_t n, int world)
{
    const size_t per = (n + (size_t) world - 1) / (size_t) world;        // rows of rank 0's shard (the longest)
    for (size_t idx = (size_t) blockIdx.x * blockDim.x + threadIdx.x; idx < n * 8; idx += (size_t) gridDim.x * blockDim.x) {
        const size_t i = idx >> 3, slot = idx & 7;
        const size_t r = i % (size_t) world, j = i / (size_t) world;
        // shards are stored rank after rank; ranks < n % world (or all, if world divides n) hold `per` rows, the others per - 1
        const size_t rem = n % (size_t) world;
        const size_t off = rem == 0 ? r * per : (r <= rem ? r * per : rem * per + (r - rem) * (per - 1));
        table[idx] = blocks[(off + j) * 8 + slot];
    }
}

}  // namespace

extern "C" int rimphony_rccl_available(void) { return rccl_ready() == RIMPHONY_OK ? 1 : 0; }

extern "C" int rimphony_rccl_unique_id(void *id128)
{
    if (!id128) return RIMPHONY_EINVAL;
    const int rc = rccl_ready();
    if (rc) return rc;
    rim_clear_last_error();
    RCCL_TRY(g_rccl.get_unique_id((RcclUniqueId *) id128));
    return RIMPHONY_OK;
}

extern "C" int rimphony_rccl_comm_create(rimphony_ctx *ctx, int rank, int world, const void *id128, void **comm)
{
    if (!ctx || !id128 || !comm || world < 1 || rank < 0 || rank >= world) return RIMPHONY_EINVAL;
    const int rc = rccl_ready();
    if (rc) return rc;
    rim_clear_last_error();
    int dev = -1;
    const int rcd = rimphony_ctx_device(ctx, &dev);
    if (rcd) return rcd;
    HIP_TRY(hipSetDevice(dev));
    RcclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    *comm = nullptr;
    RCCL_TRY(g_rccl.comm_init_rank(comm, world, id, rank));
    return RIMPHONY_OK;
}

extern "C" int rimphony_rccl_comm_destroy(void *comm)
{
    if (!comm) return RIMPHONY_OK;
    const int rc = rccl_ready();
    if (rc) return rc;
    RCCL_TRY(g_rccl.comm_destroy(comm));
    return RIMPHONY_OK;
}

extern "C" int rimphony_rccl_gather_table(rimphony_ctx *ctx, void *comm, int rank, int world, int root, size_t n_total,
                                          const double *d_shard, double *d_table, double *d_scratch, void *stream)
{
    if (!ctx || !comm || world < 1 || rank < 0 || rank >= world || root < 0 || root >= world) return RIMPHONY_EINVAL;
    if (n_total == 0) return RIMPHONY_OK;
    const size_t mine = shard_rows(n_total, rank, world);
    if (mine && !d_shard) return RIMPHONY_EINVAL;
    if (rank == root && !d_table) return RIMPHONY_EINVAL;
    const int rc = rccl_ready();
    if (rc) return rc;
    rim_clear_last_error();
    int dev = -1;
    const int rcd = rimphony_ctx_device(ctx, &dev);
    if (rcd) return rcd;
    HIP_TRY(hipSetDevice(dev));
    hipStream_t st = (hipStream_t) stream;
    double *blocks = d_scratch;
    bool own_scratch = false;
    if (rank == root && !blocks) {
        // no scratch from the caller: a private one, and the call then ends synchronised (documented in the header)
        if (hipMalloc(&blocks, n_total * 8 * sizeof(double)) != hipSuccess) {
            rim_set_last_error("rimphony_rccl_gather_table", "scratch allocation failed");
            return RIMPHONY_ENOMEM;
        }
        own_scratch = true;
    }
    // one group: every rank sends its shard to the root (the root to itself as well, so that a world of one runs the
    // same RCCL calls), the root posts one receive per rank into that rank's block
    int result = RIMPHONY_OK;
    int e = g_rccl.group_start();
    if (!e && mine) e = g_rccl.send(d_shard, mine * 8, /* ncclFloat64 */ 8, root, comm, st);
    if (!e && rank == root) {
        size_t off = 0;
        for (int r = 0; r < world && !e; r++) {
            const size_t m = shard_rows(n_total, r, world);
            if (m) e = g_rccl.recv(blocks + off * 8, m * 8, 8, r, comm, st);
            off += m;
        }
    }
    const int e_end = g_rccl.group_end();
    if (!e) e = e_end;
    if (e) { rim_set_last_error("rimphony_rccl_gather_table", g_rccl.errstr(e)); result = RIMPHONY_ERCCL; }
    if (!result && rank == root) {
        const size_t total = n_total * 8;
        const unsigned blocks_n = (unsigned) ((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
        hipLaunchKernelGGL(uninterleave_kernel, dim3(blocks_n), dim3(256), 0, st, blocks, d_table, n_total, world);
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) { rim_set_last_error("uninterleave_kernel", hipGetErrorString(le)); result = RIMPHONY_EHIP; }
    }
    if (own_scratch) {
        const hipError_t se = hipStreamSynchronize(st);
        (void) hipFree(blocks);
        if (se != hipSuccess && !result) { rim_set_last_error("hipStreamSynchronize", hipGetErrorString(se)); result = RIMPHONY_EHIP; }
    }
    return result;
}
