// Gradient-bucket all-reduce entry point of the C ABI: sum-all-reduce of one contiguous range of
// the flat gradient buffer over RCCL on the caller's side stream.  What the reference gets from HF Trainer -> accelerate ->
// DistributedDataParallel's bucketed NCCL all-reduce (TF:trainer.py:720-737); here a bucket is a range of ONE flat fp32
// buffer (params.FlatStore), so the call is a single ncclAllReduce in place.
//
// RCCL is resolved at run time (dlopen of the library the process already has - PyTorch-ROCm loads librccl - or of the
// system one), so libspeechmix_hip.so carries no link-time dependency on it and still loads on hosts without RCCL.
// The Python host (speechmix_amd/dist.py) issues the same collective through torch.distributed, because the communicator
// there is created and owned by torch's process group and cannot be handed out as an ncclComm_t; a C / C++ host that
// creates its own communicator (ncclCommInitRank) calls this.
#include <dlfcn.h>
#include "smx_common.h"

namespace {
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
nccl_allreduce_fn resolve_allreduce() {
    static nccl_allreduce_fn fn = nullptr;
    static bool tried = false;
    if (tried) return fn;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (int pass = 0; pass < 2 && !fn; ++pass)
        for (const char* n : names) {
            void* h = dlopen(n, RTLD_NOW | (pass == 0 ? RTLD_NOLOAD : 0));      // first: whatever the process already loaded
            if (!h) continue;
            fn = reinterpret_cast<nccl_allreduce_fn>(dlsym(h, "ncclAllReduce"));
            if (fn) break;
        }
    return fn;
}
}  // namespace

// comm: ncclComm_t of the caller.  buf: device pointer, reduced in place (sum).  n: elements.  dtype: SMX_F32 / SMX_BF16.
// Returns SMX_OK, SMX_EINVAL, SMX_ENOSYS when no RCCL can be loaded, or RCCL's ncclResult_t (> 0).
extern "C" int smx_allreduce_bucket(void* comm, void* buf, size_t n, int dtype, hipStream_t stream) {
    if (!comm || !buf) return SMX_EINVAL;
    if (n == 0) return SMX_OK;
    int nccl_type;
    if (dtype == SMX_F32) nccl_type = 7;            // ncclFloat32
    else if (dtype == SMX_BF16) nccl_type = 9;      // ncclBfloat16
    else return SMX_EINVAL;
    nccl_allreduce_fn fn = resolve_allreduce();
    if (!fn) return SMX_ENOSYS;
    return fn(buf, buf, n, nccl_type, /*ncclSum*/ 0, comm, stream);
}
