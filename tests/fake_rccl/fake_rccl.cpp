// fake_rccl.cpp — TEST INFRASTRUCTURE (never linked into the product): a stand-in for librccl.so that implements the seven entry points
// rt_multi.cpp uses with plain HIP copies, so that the multi-device context's PART logic (one part per device: its own stream, staging,
// communicator index, peer ranks) can be exercised on a one-GPU box.  Real RCCL refuses two ranks on one device; this fake does not.
// Loaded through RT_RCCL_LIB together with RT_MULTI_SPLIT_PARTS=1 (rt_capi.hip), see tests/test_gpu_parity.py.
//
// Semantics kept from RCCL: sends and receives are only legal inside a group; at ncclGroupEnd every send must find its receive
// (receiver's communicator rank == send.peer, recv.peer == sender's rank, same byte count, first come first matched) or the group fails;
// the data lands in the receiver's stream order, after the sender's stream has reached the send; the sender's stream does not run
// past the send before the data has been taken.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {
struct Comm { int rank, nranks, device; };
struct Op { bool send; const void* src; void* dst; size_t bytes; int peer; Comm* comm; hipStream_t stream; bool matched; };
std::vector<Op> g_ops;
int g_depth = 0;
std::string g_log;
void logf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void logf(const char* fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_log += buf;
}

int flush_group()
{
    int rc = 0;
    for (Op& s : g_ops) {
        if (!s.send) continue;
        Op* r = nullptr;
        for (Op& c : g_ops)
            if (!c.send && !c.matched && c.comm->rank == s.peer && c.peer == s.comm->rank && c.bytes == s.bytes) { r = &c; break; }
        if (!r) { rc = 5; logf("unmatched send rank %d -> %d (%zu B)\n", s.comm->rank, s.peer, s.bytes); continue; }
        r->matched = s.matched = true;
        hipEvent_t ready, taken;
        if (hipSetDevice(s.comm->device) != hipSuccess || hipEventCreateWithFlags(&ready, hipEventDisableTiming) != hipSuccess ||
            hipEventRecord(ready, s.stream) != hipSuccess) { rc = 1; continue; }
        if (hipSetDevice(r->comm->device) != hipSuccess || hipStreamWaitEvent(r->stream, ready, 0) != hipSuccess ||
            hipMemcpyAsync(r->dst, s.src, s.bytes, hipMemcpyDeviceToDevice, r->stream) != hipSuccess ||
            hipEventCreateWithFlags(&taken, hipEventDisableTiming) != hipSuccess || hipEventRecord(taken, r->stream) != hipSuccess ||
            hipStreamWaitEvent(s.stream, taken, 0) != hipSuccess) { rc = 1; continue; }
        (void)hipEventDestroy(ready);  // destruction is deferred by the runtime until the recorded work has completed
        (void)hipEventDestroy(taken);
    }
    for (Op& c : g_ops)
        if (!c.send && !c.matched) { rc = 5; logf("unmatched recv rank %d <- %d (%zu B)\n", c.comm->rank, c.peer, c.bytes); }
    g_ops.clear();
    return rc;
}
}  // namespace

extern "C" {

int ncclCommInitAll(void** comms, int ndev, const int* devlist)
{
    logf("init %d:", ndev);
    for (int i = 0; i < ndev; ++i) {
        comms[i] = new Comm{i, ndev, devlist ? devlist[i] : i};
        logf(" %d", devlist ? devlist[i] : i);
    }
    logf("\n");
    return 0;
}

int ncclCommDestroy(void* comm)
{
    delete static_cast<Comm*>(comm);
    return 0;
}

int ncclGroupStart()
{
    if (g_depth++ == 0) logf("group\n");
    return 0;
}

int ncclGroupEnd()
{
    if (g_depth <= 0) return 5;
    if (--g_depth > 0) return 0;
    const int rc = flush_group();
    logf("end %d\n", rc);
    return rc;
}

int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (g_depth <= 0 || dtype != 1 || !c || peer < 0 || peer >= c->nranks) return 5;
    logf("send %d->%d %zu\n", c->rank, peer, count);
    g_ops.push_back(Op{true, buf, nullptr, count, peer, c, stream, false});
    return 0;
}

int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream)
{
    Comm* c = static_cast<Comm*>(comm);
    if (g_depth <= 0 || dtype != 1 || !c || peer < 0 || peer >= c->nranks) return 5;
    logf("recv %d<-%d %zu\n", c->rank, peer, count);
    g_ops.push_back(Op{false, nullptr, buf, count, peer, c, stream, false});
    return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "no error" : (rc == 5 ? "invalid usage (fake RCCL)" : "unhandled HIP error (fake RCCL)"); }

// test hooks: the call log since the last fake_rccl_log_reset (copied into buf, NUL-terminated; returns the full length)
size_t fake_rccl_log(char* buf, size_t cap)
{
    if (buf && cap) {
        const size_t n = g_log.size() < cap - 1 ? g_log.size() : cap - 1;
        std::memcpy(buf, g_log.data(), n);
        buf[n] = 0;
    }
    return g_log.size();
}
void fake_rccl_log_reset() { g_log.clear(); }

}  // extern "C"
