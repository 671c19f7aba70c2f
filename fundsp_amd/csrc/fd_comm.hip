// fd_comm.hip -- the one exchange step of the path: the stereo mix-down across GPUs.
//
// Voices shard across GPUs with no data-path collective; what remains is ONE all-reduce(sum, f32) of the per-GPU
// partial mixes [2][frames] (8 * frames bytes per GPU: 512 B for a 64-frame block, 384 KB for a second) over
// RCCL / xGMI.  It is latency-bound, so it is issued once per launch on the communicator's own SIDE stream, ordered
// after the mix kernel through an event, and the caller's render stream is free to start the next launch at once;
// whoever consumes the mix waits on the completion event (fdsp_comm_wait).
//
// Two ways to build the communicator, both plain RCCL underneath (librccl.so is a link-time dependency of
// libfundsp_hip.so):
//   fdsp_comm_create_local(n, devices)          one process driving n GPUs (ncclCommInitAll): slot k = devices[k]
//   fdsp_comm_unique_id + fdsp_comm_create_rank one process per GPU (torchrun, MPI ...): the id travels out of band
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <vector>

#include "../../include/fundsp_hip.h"
#include <hip/hip_runtime.h>

namespace fd {
int api_fail(int code, const std::string& msg);  // fd_capi.hip: records fdsp_last_error() for this thread
}

struct fdsp_comm {
    struct Slot {
        int device = -1;
        ncclComm_t comm = nullptr;
        hipStream_t side = nullptr;     // the all-reduce runs here, next to the render streams
        hipEvent_t ready = nullptr;     // recorded on the producer's stream: the partial mix is complete
        hipEvent_t done = nullptr;      // recorded on the side stream: the summed mix is in place
        bool pending = false;
    };
    std::vector<Slot> slots;  // the ranks THIS process owns (n for a local communicator, 1 for a rank communicator)
    int nranks = 0;
};

namespace {
struct DevGuard {
    int prev = -1;
    bool switched = false;
    explicit DevGuard(int dev) {
        if (dev >= 0 && hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DevGuard() {
        if (switched) hipSetDevice(prev);
    }
};
#define NCCLCHK(expr)                                                                                              \
    do {                                                                                                           \
        ncclResult_t r_ = (expr);                                                                                  \
        if (r_ != ncclSuccess) return fd::api_fail(FDSP_EDEVICE, std::string(#expr) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define HIPCHK2(expr)                                                                                            \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return fd::api_fail(FDSP_EDEVICE, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

int slot_setup(fdsp_comm::Slot& s) {
    DevGuard g(s.device);
    HIPCHK2(hipStreamCreateWithFlags(&s.side, hipStreamNonBlocking));
    HIPCHK2(hipEventCreateWithFlags(&s.ready, hipEventDisableTiming));
    HIPCHK2(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    return FDSP_OK;
}
}  // namespace

extern "C" {

int fdsp_comm_create_local(int n, const int* devices, fdsp_comm** out) {
    if (!out) return fd::api_fail(FDSP_EINVAL, "out is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fd::api_fail(FDSP_EDEVICE, "no HIP device available");
    if (n < 1 || n > ndev) return fd::api_fail(FDSP_EINVAL, "need 1 <= n <= device count");
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        devs[i] = devices ? devices[i] : i;
        if (devs[i] < 0 || devs[i] >= ndev) return fd::api_fail(FDSP_EINVAL, "device index out of range");
    }
    std::vector<ncclComm_t> comms(n);
    NCCLCHK(ncclCommInitAll(comms.data(), n, devs.data()));
    fdsp_comm* c = new fdsp_comm();
    c->nranks = n;
    c->slots.resize(n);
    for (int i = 0; i < n; i++) {  // every ncclComm_t has its slot BEFORE anything can fail: fdsp_comm_destroy frees them all
        c->slots[i].device = devs[i];
        c->slots[i].comm = comms[i];
    }
    for (int i = 0; i < n; i++)
        if (int rc = slot_setup(c->slots[i])) {
            fdsp_comm_destroy(c);
            return rc;
        }
    *out = c;
    return FDSP_OK;
}

int fdsp_comm_unique_id(void* id128) {
    if (!id128) return fd::api_fail(FDSP_EINVAL, "id buffer is NULL");
    static_assert(NCCL_UNIQUE_ID_BYTES == FDSP_COMM_ID_BYTES, "id size");
    ncclUniqueId id;
    NCCLCHK(ncclGetUniqueId(&id));
    std::memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
    return FDSP_OK;
}

int fdsp_comm_create_rank(const void* id128, int nranks, int rank, int device, fdsp_comm** out) {
    if (!out || !id128) return fd::api_fail(FDSP_EINVAL, "out or id is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fd::api_fail(FDSP_EDEVICE, "no HIP device available");
    if (nranks < 1 || rank < 0 || rank >= nranks) return fd::api_fail(FDSP_EINVAL, "bad rank / nranks");
    if (device < 0 && hipGetDevice(&device) != hipSuccess) return fd::api_fail(FDSP_EDEVICE, "no current device");
    if (device >= ndev) return fd::api_fail(FDSP_EINVAL, "device index out of range");
    DevGuard g(device);
    ncclUniqueId id;
    std::memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    ncclComm_t comm = nullptr;
    NCCLCHK(ncclCommInitRank(&comm, nranks, id, rank));
    fdsp_comm* c = new fdsp_comm();
    c->nranks = nranks;
    c->slots.resize(1);
    c->slots[0].device = device;
    c->slots[0].comm = comm;
    if (int rc = slot_setup(c->slots[0])) {
        fdsp_comm_destroy(c);
        return rc;
    }
    *out = c;
    return FDSP_OK;
}

void fdsp_comm_destroy(fdsp_comm* c) {
    if (!c) return;
    for (auto& s : c->slots) {
        DevGuard g(s.device);
        if (s.side) hipStreamSynchronize(s.side);
        if (s.comm) ncclCommDestroy(s.comm);
        if (s.ready) hipEventDestroy(s.ready);
        if (s.done) hipEventDestroy(s.done);
        if (s.side) hipStreamDestroy(s.side);
    }
    delete c;
}

int fdsp_comm_ranks(const fdsp_comm* c) { return c ? c->nranks : FDSP_EINVAL; }
int fdsp_comm_local_slots(const fdsp_comm* c) { return c ? (int)c->slots.size() : FDSP_EINVAL; }
int fdsp_comm_device(const fdsp_comm* c, int slot) {
    return (c && slot >= 0 && slot < (int)c->slots.size()) ? c->slots[slot].device : FDSP_EINVAL;
}

// one slot: thread-per-GPU hosts and one-process-per-GPU hosts call this from each owner
int fdsp_mix_allreduce(fdsp_comm* c, int slot, float* d_mix, size_t count, void* after_stream) {
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fd::api_fail(FDSP_EINVAL, "bad communicator or slot");
    if (!d_mix) return fd::api_fail(FDSP_EINVAL, "d_mix is NULL");
    if (count == 0) return FDSP_OK;
    fdsp_comm::Slot& s = c->slots[slot];
    DevGuard g(s.device);
    HIPCHK2(hipEventRecord(s.ready, (hipStream_t)after_stream));
    HIPCHK2(hipStreamWaitEvent(s.side, s.ready, 0));
    NCCLCHK(ncclAllReduce(d_mix, d_mix, count, ncclFloat32, ncclSum, s.comm, s.side));
    HIPCHK2(hipEventRecord(s.done, s.side));
    s.pending = true;
    return FDSP_OK;
}

// all local slots at once from ONE host thread (a local communicator needs the calls grouped, or they would wait for
// each other): d_mix[k] / after_streams[k] belong to slot k
int fdsp_mix_allreduce_all(fdsp_comm* c, float* const* d_mix, size_t count, void* const* after_streams) {
    if (!c || !d_mix) return fd::api_fail(FDSP_EINVAL, "bad communicator or buffers");
    if (count == 0) return FDSP_OK;
    const int n = (int)c->slots.size();
    for (int k = 0; k < n; k++) {
        if (!d_mix[k]) return fd::api_fail(FDSP_EINVAL, "d_mix[k] is NULL");
        fdsp_comm::Slot& s = c->slots[k];
        DevGuard g(s.device);
        HIPCHK2(hipEventRecord(s.ready, after_streams ? (hipStream_t)after_streams[k] : nullptr));
        HIPCHK2(hipStreamWaitEvent(s.side, s.ready, 0));
    }
    NCCLCHK(ncclGroupStart());
    for (int k = 0; k < n; k++) {
        fdsp_comm::Slot& s = c->slots[k];
        ncclResult_t r = ncclAllReduce(d_mix[k], d_mix[k], count, ncclFloat32, ncclSum, s.comm, s.side);
        if (r != ncclSuccess) {
            ncclGroupEnd();
            return fd::api_fail(FDSP_EDEVICE, std::string("ncclAllReduce: ") + ncclGetErrorString(r));
        }
    }
    NCCLCHK(ncclGroupEnd());
    for (int k = 0; k < n; k++) {
        fdsp_comm::Slot& s = c->slots[k];
        DevGuard g(s.device);
        HIPCHK2(hipEventRecord(s.done, s.side));
        s.pending = true;
    }
    return FDSP_OK;
}

// order `stream` behind the slot's last all-reduce; stream == NULL blocks the host until it is complete
int fdsp_comm_wait(fdsp_comm* c, int slot, void* stream) {
    if (!c || slot < 0 || slot >= (int)c->slots.size()) return fd::api_fail(FDSP_EINVAL, "bad communicator or slot");
    fdsp_comm::Slot& s = c->slots[slot];
    if (!s.pending) return FDSP_OK;
    DevGuard g(s.device);
    if (stream) HIPCHK2(hipStreamWaitEvent((hipStream_t)stream, s.done, 0));
    else HIPCHK2(hipEventSynchronize(s.done));
    return FDSP_OK;
}

}  // extern "C"
