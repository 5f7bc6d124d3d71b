// The one collective of the path for a C caller: RCCL all-gather of the per-clip outputs over xGMI.
// Clips are independent in eval mode (convnext.py:219,305), every rank holds a full weight replica and scores its shard;
// what crosses the links is logits / probs / scene rows only (135 KB per rank at 64 clips) -- frame embeddings stay sharded.
// The reference has no inference-time collective (its only parallelism is training DDP, main.py:641,992-997); the Python host
// reaches the same collective through torch.distributed (parallel.py).
//
// librccl.so is opened on first use (dlopen): libacx.so itself has no link-time dependency on it, a single-GPU process never
// loads it.  No CUDA-compat layer: these are RCCL's own entry points (the nccl* names ARE RCCL's API).
#include <dlfcn.h>
#include <link.h>
#include <cstdio>
#include <cstring>

#include "acx_internal.h"

namespace acx {

namespace {
typedef struct { char internal[ACX_COMM_ID_BYTES]; } rccl_unique_id;     // ncclUniqueId: 128 opaque bytes (rccl.h)
typedef void* rccl_comm;
struct Rccl {
    void* so = nullptr;
    int (*get_unique_id)(rccl_unique_id*) = nullptr;
    int (*comm_init_rank)(rccl_comm*, int, rccl_unique_id, int) = nullptr;
    int (*comm_destroy)(rccl_comm) = nullptr;
    int (*all_gather)(const void*, void*, size_t, int /*ncclDataType_t*/, rccl_comm, hipStream_t) = nullptr;
    const char* (*error_string)(int) = nullptr;
};
int load_rccl(Rccl** out) {
    static Rccl r;
    static std::once_flag once;
    static const char* err = nullptr;
    std::call_once(once, [] {
        // One RCCL per process: if a copy is already mapped (torch ships its own next to libtorch and torch.distributed has
        // loaded it), bind to THAT one -- by soname first, then by walking the loaded objects -- and open a new one only when
        // none is (a plain C caller: /opt/rocm/lib).  Two instances would each keep their own communicator tables and
        // bootstrap state.
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_NOLOAD);
            if (r.so) break;
        }
        if (!r.so) {
            static char found[1024];
            found[0] = 0;
            dl_iterate_phdr([](struct dl_phdr_info* info, size_t, void*) -> int {
                if (info->dlpi_name && std::strstr(info->dlpi_name, "librccl.so")) { std::snprintf(found, sizeof found, "%s", info->dlpi_name); return 1; }
                return 0;
            }, nullptr);
            if (found[0]) r.so = dlopen(found, RTLD_NOW | RTLD_NOLOAD);
        }
        if (!r.so)
            for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
                r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (r.so) break;
            }
        if (!r.so) { err = "librccl.so not found (dlopen)"; return; }
        r.get_unique_id = reinterpret_cast<decltype(r.get_unique_id)>(dlsym(r.so, "ncclGetUniqueId"));
        r.comm_init_rank = reinterpret_cast<decltype(r.comm_init_rank)>(dlsym(r.so, "ncclCommInitRank"));
        r.comm_destroy = reinterpret_cast<decltype(r.comm_destroy)>(dlsym(r.so, "ncclCommDestroy"));
        r.all_gather = reinterpret_cast<decltype(r.all_gather)>(dlsym(r.so, "ncclAllGather"));
        r.error_string = reinterpret_cast<decltype(r.error_string)>(dlsym(r.so, "ncclGetErrorString"));
        if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.all_gather) err = "librccl.so lacks an expected entry point";
    });
    if (err) ACX_FAIL(ACX_ERR_UNSUPPORTED, "RCCL: %s", err);
    *out = &r;
    return ACX_OK;
}
#define ACX_RCCL(r_, expr_)                                                                                  \
    do {                                                                                                     \
        const int e_ = (expr_);                                                                              \
        if (e_ != 0) ACX_FAIL(ACX_ERR_HIP, "%s failed: %s", #expr_, (r_)->error_string ? (r_)->error_string(e_) : "RCCL error"); \
    } while (0)
}  // namespace

void comm_release(acx_ctx* c) {
    if (!c || !c->comm) return;
    Rccl* r = nullptr;
    if (load_rccl(&r) == ACX_OK) (void)r->comm_destroy(reinterpret_cast<rccl_comm>(c->comm));
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
}

}  // namespace acx

using namespace acx;

extern "C" {

int acx_comm_unique_id(void* id_out) {
    if (!id_out) ACX_FAIL(ACX_ERR_ARG, "acx_comm_unique_id: null pointer");
    Rccl* r = nullptr;
    ACX_TRY(load_rccl(&r));
    rccl_unique_id id;
    ACX_RCCL(r, r->get_unique_id(&id));
    std::memcpy(id_out, &id, sizeof id);
    return ACX_OK;
}

int acx_comm_init(acx_ctx* c, int rank, int world, const void* unique_id) {
    if (!c || !unique_id || world < 1 || rank < 0 || rank >= world) ACX_FAIL(ACX_ERR_ARG, "acx_comm_init: bad argument");
    Rccl* r = nullptr;
    ACX_TRY(load_rccl(&r));
    ACX_HIP(hipSetDevice(c->device));
    comm_release(c);
    rccl_unique_id id;
    std::memcpy(&id, unique_id, sizeof id);
    rccl_comm comm = nullptr;
    ACX_RCCL(r, r->comm_init_rank(&comm, world, id, rank));
    c->comm = comm; c->comm_rank = rank; c->comm_world = world;
    return ACX_OK;
}

int acx_allgather(acx_ctx* c, const void* send, void* recv, size_t bytes_per_rank, void* stream) {
    if (!c || !send || !recv) ACX_FAIL(ACX_ERR_ARG, "acx_allgather: null pointer");
    if (!c->comm) ACX_FAIL(ACX_ERR_STATE, "acx_allgather: call acx_comm_init first");
    if (bytes_per_rank == 0) return ACX_OK;
    Rccl* r = nullptr;
    ACX_TRY(load_rccl(&r));
    // bytes as ncclInt8 (= 0): the payload is opaque rows (logits, probabilities, scene embeddings)
    ACX_RCCL(r, r->all_gather(send, recv, bytes_per_rank, 0, reinterpret_cast<rccl_comm>(c->comm), (hipStream_t)stream));
    return ACX_OK;
}

int acx_comm_info(const acx_ctx* c, int* rank, int* world) {
    if (!c) ACX_FAIL(ACX_ERR_ARG, "null context");
    if (rank) *rank = c->comm ? c->comm_rank : 0;
    if (world) *world = c->comm ? c->comm_world : 1;
    return ACX_OK;
}

}  // extern "C"
