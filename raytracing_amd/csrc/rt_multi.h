// rt_multi.h — the RCCL side of a multi-device context (rt_ctx_create_multi): communicators over the distinct devices of ONE process
// and the per-frame gather of the row bands into the root device's frame as ONE group of ncclSend / ncclRecv (xGMI between devices,
// a local copy for a band that already sits on the root device).  RCCL is loaded at run time (dlopen) the first time a context with
// more than one band is created, so that single-device users of the library never need it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <string>
#include <vector>

namespace rtamd {

// One band transfer of a frame: `bytes` from `src` on device part `src_part` (enqueued on that part's stream) into `dst` on the
// root device (part 0, enqueued on the root stream of the call).
struct BandXfer {
    int src_part;
    const void* src;
    void* dst;
    size_t bytes;
    hipStream_t src_stream;
};

class BandGather {
public:
    // `devices`: the distinct HIP devices, devices[0] = root.  Returns "" or the error text (RCCL missing, ncclCommInitAll failed ...).
    std::string init(const std::vector<int>& devices);
    // One ncclGroupStart ... ncclGroupEnd holding, per transfer, the send on the source part's communicator and the matching receive on
    // the root's.  Transfers from one part are matched in order (RCCL's rule for several sends to one peer inside a group).
    std::string gather(const std::vector<BandXfer>& xfers, hipStream_t root_stream);
    void destroy();
    bool ready() const { return !comms_.empty(); }
    const std::string& library_path() const { return path_; }

private:
    std::vector<void*> comms_;  // ncclComm_t per part
    std::string path_;
};

}  // namespace rtamd
