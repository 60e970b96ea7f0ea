// rt_multi.cpp — see rt_multi.h.  Host code only; RCCL through dlopen (function pointers), no link-time dependency.
#include "rt_multi.h"

#include <dlfcn.h>

#include <cstdlib>

namespace rtamd {
namespace {

// rccl.h, the handful of entry points used here (ncclResult_t and ncclDataType_t are ints; ncclUint8 = 1; ncclComm_t is a pointer)
struct RcclApi {
    void* handle = nullptr;
    std::string path;
    int (*CommInitAll)(void** comms, int ndev, const int* devlist) = nullptr;
    int (*CommDestroy)(void* comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) = nullptr;
    int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t stream) = nullptr;
    const char* (*GetErrorString)(int rc) = nullptr;
};
constexpr int kNcclUint8 = 1;

RcclApi g_api;  // loaded once per process (first multi-band context); creation of contexts is not thread-safe, like the rest of a ctx

std::string load_api()
{
    if (g_api.handle) return "";
    // $RT_RCCL_LIB (an explicit choice) first; else an RCCL that is already mapped (a host that also uses torch.distributed must end up
    // with ONE RCCL in the process); else the loader's search path and the ROCm default location.
    const char* env = std::getenv("RT_RCCL_LIB");
    void* h = nullptr;
    std::string where;
    if (env && *env) {
        if (!(h = dlopen(env, RTLD_NOW | RTLD_GLOBAL))) {
            const char* e = dlerror();
            return std::string("RT_RCCL_LIB=") + env + " could not be loaded: " + (e ? e : "");
        }
        where = env;
    }
    const char* names[] = {"librccl.so", "librccl.so.1"};
    for (const char* n : names)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL))) where = std::string(n) + " (already loaded)";
    const char* paths[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : paths)
        if (!h && (h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) where = n;
    if (!h) {
        const char* e = dlerror();
        return std::string("RCCL not found (librccl.so; set RT_RCCL_LIB): ") + (e ? e : "");
    }
    RcclApi a;
    a.handle = h;
    a.path = where;
    a.CommInitAll = reinterpret_cast<decltype(a.CommInitAll)>(dlsym(h, "ncclCommInitAll"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    a.GroupStart = reinterpret_cast<decltype(a.GroupStart)>(dlsym(h, "ncclGroupStart"));
    a.GroupEnd = reinterpret_cast<decltype(a.GroupEnd)>(dlsym(h, "ncclGroupEnd"));
    a.Send = reinterpret_cast<decltype(a.Send)>(dlsym(h, "ncclSend"));
    a.Recv = reinterpret_cast<decltype(a.Recv)>(dlsym(h, "ncclRecv"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (!a.CommInitAll || !a.CommDestroy || !a.GroupStart || !a.GroupEnd || !a.Send || !a.Recv || !a.GetErrorString)
        return "the RCCL at " + where + " lacks one of ncclCommInitAll/ncclCommDestroy/ncclGroupStart/ncclGroupEnd/ncclSend/ncclRecv";
    g_api = a;
    return "";
}

std::string nccl_error(const char* what, int rc) { return std::string(what) + ": " + g_api.GetErrorString(rc); }

}  // namespace

std::string BandGather::init(const std::vector<int>& devices)
{
    if (devices.empty()) return "no devices";
    std::string msg = load_api();
    if (!msg.empty()) return msg;
    path_ = g_api.path;
    int prev = -1;
    (void)hipGetDevice(&prev);
    comms_.assign(devices.size(), nullptr);
    const int rc = g_api.CommInitAll(comms_.data(), (int)devices.size(), devices.data());
    if (prev >= 0) (void)hipSetDevice(prev);  // ncclCommInitAll walks the devices
    if (rc != 0) {
        comms_.clear();
        return nccl_error("ncclCommInitAll", rc);
    }
    return "";
}

std::string BandGather::gather(const std::vector<BandXfer>& xfers, hipStream_t root_stream)
{
    if (xfers.empty()) return "";
    if (comms_.empty()) return "the band gather was not initialised";
    for (const BandXfer& x : xfers)
        if (x.src_part < 0 || (size_t)x.src_part >= comms_.size() || !x.src || !x.dst) return "bad band transfer";
    int prev = -1;
    (void)hipGetDevice(&prev);
    std::string err;
    int rc = g_api.GroupStart();
    if (rc != 0) return nccl_error("ncclGroupStart", rc);
    for (const BandXfer& x : xfers) {
        if (x.bytes == 0) continue;
        if ((rc = g_api.Recv(x.dst, x.bytes, kNcclUint8, x.src_part, comms_[0], root_stream)) != 0) { err = nccl_error("ncclRecv", rc); break; }
        if ((rc = g_api.Send(x.src, x.bytes, kNcclUint8, 0, comms_[(size_t)x.src_part], x.src_stream)) != 0) { err = nccl_error("ncclSend", rc); break; }
    }
    rc = g_api.GroupEnd();  // always closes the group; the first error is the one reported
    if (err.empty() && rc != 0) err = nccl_error("ncclGroupEnd", rc);
    if (prev >= 0) (void)hipSetDevice(prev);  // a group over several communicators may leave another device current
    return err;
}

void BandGather::destroy()
{
    for (void* c : comms_)
        if (c) (void)g_api.CommDestroy(c);
    comms_.clear();
}

}  // namespace rtamd
