// exchange.hip -- the path's ONE cross-GPU step, in C: the int32 all-reduce of the split conferences' partial sums
// (what the loop of mixer_process, src/audiofilters/audiomixer.c:304-314, becomes when a conference's members
// live on several GPUs: mi_mixer_partial_sum -> mi_exchange_allreduce_i32 -> mi_mixer_finalize).
//
// Straight on RCCL (librccl, xGMI between the GPUs of a node): one communicator rank per context.  The collective is
// enqueued on the CONTEXT'S OWN stream, so it is ordered after the partial sums and before the finalize by stream order --
// no second stream, no events, no host synchronisation.  The ranks may be threads of one process (a mediastreamer2
// process runs one ticker thread per conference, src/voip/audioconference.c:72) or processes; the 128-byte id made by
// one of them travels by whatever means the host has (shared memory, a socket, torch's store in bench.py).
//
// librccl is opened on first use (dlopen): a host that never splits a conference does not load it, and a box without
// the LIBRARY still runs everything else (the header rccl/rccl.h is needed to build: the types are RCCL's).  Failure is
// loud: every entry point returns MI_ENODEV with RCCL's -- or the loader's -- own message.
#include "common.hpp"

#include <dlfcn.h>
#include <mutex>
#include <string>
#include <rccl/rccl.h>

namespace {

struct Rccl {
	void *lib = nullptr;
	ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
	ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
	std::string why; // the loader's message, captured where dlopen failed (dlerror() is per thread and cleared by reading)
};

Rccl &rccl() {
	static Rccl r;
	static std::once_flag once;
	std::call_once(once, [] {
		const char *override_path = getenv("MSMI355X_RCCL_LIB"); // tests: a library with the same five entry points
		for (const char *name : {override_path, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			if (!name || !*name) continue;
			r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
			if (r.lib) break;
			if (const char *e = dlerror()) r.why += (r.why.empty() ? "" : "; ") + std::string(e);
		}
		if (!r.lib) return;
		r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.lib, "ncclGetUniqueId");
		r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.lib, "ncclCommInitRank");
		r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.lib, "ncclCommDestroy");
		r.AllReduce = (decltype(r.AllReduce))dlsym(r.lib, "ncclAllReduce");
		r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.lib, "ncclGetErrorString");
		r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllReduce && r.GetErrorString;
		if (!r.ok) r.why = "an ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / GetErrorString symbol is missing";
	});
	return r;
}

int rccl_missing() {
	mi::set_error("librccl could not be loaded (%s): the cross-GPU conference exchange is unavailable", rccl().why.c_str());
	return MI_ENODEV;
}

#define MI_RCCL(expr)                                                                               \
	do {                                                                                            \
		ncclResult_t r__ = (expr);                                                                  \
		if (r__ != ncclSuccess) {                                                                   \
			mi::set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, rccl().GetErrorString(r__)); \
			return MI_ENODEV;                                                                       \
		}                                                                                           \
	} while (0)

} // namespace

struct mi_exchange {
	mi_ctx *ctx = nullptr;
	ncclComm_t comm = nullptr;
	int nranks = 0, rank = 0;
};

extern "C" {

int mi_exchange_unique_id(void *id_out, size_t cap) {
	MI_CHECK_ARG(id_out && cap >= MI_EXCHANGE_ID_BYTES);
	static_assert(MI_EXCHANGE_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's");
	if (!rccl().ok) return rccl_missing();
	ncclUniqueId id;
	MI_RCCL(rccl().GetUniqueId(&id));
	memcpy(id_out, &id, sizeof(id));
	return MI_OK;
}

int mi_exchange_create(mi_ctx *ctx, int nranks, int rank, const void *unique_id, mi_exchange **out) {
	MI_CHECK_ARG(ctx && out && unique_id && nranks >= 1 && rank >= 0 && rank < nranks);
	*out = nullptr;
	if (!rccl().ok) return rccl_missing();
	if (ctx->activate() != MI_OK) return MI_ENODEV;
	ncclUniqueId id;
	memcpy(&id, unique_id, sizeof(id));
	mi_exchange *x = new mi_exchange();
	x->ctx = ctx;
	x->nranks = nranks;
	x->rank = rank;
	const ncclResult_t r = rccl().CommInitRank(&x->comm, nranks, id, rank); // returns once every rank has joined
	if (r != ncclSuccess) {
		mi::set_error("ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, nranks, ctx->device, rccl().GetErrorString(r));
		delete x;
		return MI_ENODEV;
	}
	*out = x;
	return MI_OK;
}

void mi_exchange_destroy(mi_exchange *x) {
	if (!x) return;
	(void)hipSetDevice(x->ctx->device);
	(void)hipStreamSynchronize(x->ctx->stream);
	if (x->comm) (void)rccl().CommDestroy(x->comm);
	delete x;
}

int mi_exchange_ranks(const mi_exchange *x, int *nranks, int *rank) {
	MI_CHECK_ARG(x != nullptr);
	if (nranks) *nranks = x->nranks;
	if (rank) *rank = x->rank;
	return MI_OK;
}

int mi_exchange_allreduce_i32(mi_exchange *x, int32_t *d_buf, size_t count) {
	MI_CHECK_ARG(x && d_buf && count > 0);
	if (x->ctx->activate() != MI_OK) return MI_ENODEV;
	MI_RCCL(rccl().AllReduce(d_buf, d_buf, count, ncclInt32, ncclSum, x->comm, x->ctx->stream));
	return MI_OK;
}

} // extern "C"
