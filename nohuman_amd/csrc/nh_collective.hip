// nh_collective.hip -- the one exchange step of the path: the per-device counters
// {fragments, classified, bases, table lookups} of a multi-device run are summed with ONE
// ncclAllReduce(4 x uint64, sum) over RCCL / xGMI, single process, ncclCommInitAll over the chosen
// devices.  What kraken2's three stderr summary integers (/root/reference/src/lib.rs:61-97) become
// when reads shard over several GPUs.  RCCL is bound at run time (dlopen): the library of the HIP
// runtime this process already uses is preferred, so that no second runtime is pulled in.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <link.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "nh_internal.h"
#include "nohuman_engine.h"

namespace nh {
namespace {

typedef void *ncclComm_t;
struct Rccl {
    void *h = nullptr;
    std::string path;
    int (*GetVersion)(int *) = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint64 = 5, kNcclSum = 0;  // rccl.h: ncclDataType_t / ncclRedOp_t

struct Loaded {
    std::string rccl, hip;
};
int phdr_cb(struct dl_phdr_info *info, size_t, void *data) {
    Loaded *l = (Loaded *)data;
    const char *n = info->dlpi_name ? info->dlpi_name : "";
    if (strstr(n, "librccl.so") && l->rccl.empty()) l->rccl = n;
    if (strstr(n, "libamdhip64.so") && l->hip.empty()) l->hip = n;
    return 0;
}

int load_rccl_once(Rccl &r, std::string &why) {
    Loaded l;
    dl_iterate_phdr(phdr_cb, &l);
    std::vector<std::string> cand;
    if (!l.rccl.empty()) cand.push_back(l.rccl);  // already in the process (e.g. torch's)
    if (!l.hip.empty()) {                         // the one that ships with the loaded HIP runtime
        const size_t s = l.hip.rfind('/');
        if (s != std::string::npos) {
            cand.push_back(l.hip.substr(0, s) + "/librccl.so.1");
            cand.push_back(l.hip.substr(0, s) + "/librccl.so");
        }
    }
    cand.push_back("librccl.so.1");
    cand.push_back("librccl.so");
    cand.push_back("/opt/rocm/lib/librccl.so.1");
    for (const std::string &c : cand) {
        r.h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (r.h) {
            r.path = c;
            break;
        }
    }
    if (!r.h) {
        why = "librccl.so not found";
        return -1;
    }
    bool ok = true;
    auto sym = [&](const char *n) {
        void *p = dlsym(r.h, n);
        if (!p) {
            ok = false;
            why = std::string("missing symbol ") + n;
        }
        return p;
    };
    r.GetVersion = (int (*)(int *))sym("ncclGetVersion");
    r.CommInitAll = (int (*)(ncclComm_t *, int, const int *))sym("ncclCommInitAll");
    r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
    r.AllReduce = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclAllReduce");
    r.GroupStart = (int (*)())sym("ncclGroupStart");
    r.GroupEnd = (int (*)())sym("ncclGroupEnd");
    r.GetErrorString = (const char *(*)(int))sym("ncclGetErrorString");
    return ok ? 0 : -1;
}

// RCCL is resolved once per process and stays loaded (a handle per call leaked)
struct RcclOnce {
    Rccl r;
    std::string why;
    int rc;
    RcclOnce() { rc = load_rccl_once(r, why); }
};
const RcclOnce &rccl() {
    static const RcclOnce once;
    return once;
}

}  // namespace

// rows: n_dev x 4 uint64.  d_src == NULL: row g holds the counters of device ids[g] (uploaded for the
// reduction); d_src[g] != NULL: the counters are the four uint64 the classify kernels of the run have
// been adding to in device ids[g]'s HBM -- they are reduced where they lie (out of place into a scratch
// row per device), nothing is uploaded.  On success every row holds the sum.
int allreduce_counters(const int *ids, int n_dev, uint64_t *rows, std::string &backend, const uint64_t *const *d_src) {
    if (!ids || !rows || n_dev <= 0) return set_error(NH_EINVAL, "allreduce_counters: bad argument");
    const RcclOnce &once = rccl();
    if (once.rc != 0) return set_error(NH_EDEVICE, "RCCL unavailable: %s", once.why.c_str());
    const Rccl &r = once.r;
    int version = 0;
    (void)r.GetVersion(&version);
    std::vector<ncclComm_t> comms((size_t)n_dev, nullptr);
    std::vector<int> hip_ids((size_t)n_dev);  // (RCCL takes HIP ordinals; `ids` are this library's logical devices)
    for (int g = 0; g < n_dev; g++) hip_ids[(size_t)g] = dev_phys(ids[g]);
    int rc = r.CommInitAll(comms.data(), n_dev, hip_ids.data());
    if (rc != 0) return set_error(NH_EDEVICE, "ncclCommInitAll over %d device(s): %s", n_dev, r.GetErrorString(rc));
    std::vector<uint64_t *> d((size_t)n_dev, nullptr);
    std::vector<hipStream_t> st((size_t)n_dev, nullptr);
    hipError_t he = hipSuccess;
    for (int g = 0; g < n_dev && he == hipSuccess; g++) {
        he = dev_set(ids[g]);
        if (he == hipSuccess) he = hipStreamCreateWithFlags(&st[g], hipStreamNonBlocking);
        if (he == hipSuccess) he = dev_malloc((void **)&d[g], 4 * sizeof(uint64_t));
        if (he == hipSuccess && !(d_src && d_src[g]))
            he = hipMemcpyAsync(d[g], rows + 4 * g, 4 * sizeof(uint64_t), hipMemcpyHostToDevice, st[g]);
    }
    int nrc = 0;
    if (he == hipSuccess) {
        nrc = r.GroupStart();
        for (int g = 0; g < n_dev && nrc == 0; g++) {
            he = dev_set(ids[g]);
            if (he != hipSuccess) break;
            const void *send = (d_src && d_src[g]) ? (const void *)d_src[g] : (const void *)d[g];
            nrc = r.AllReduce(send, d[g], 4, kNcclUint64, kNcclSum, comms[g], st[g]);
        }
        const int erc = r.GroupEnd();
        if (nrc == 0) nrc = erc;
    }
    for (int g = 0; g < n_dev && he == hipSuccess && nrc == 0; g++) {
        he = dev_set(ids[g]);
        if (he == hipSuccess) he = hipMemcpyAsync(rows + 4 * g, d[g], 4 * sizeof(uint64_t), hipMemcpyDeviceToHost, st[g]);
        if (he == hipSuccess) he = hipStreamSynchronize(st[g]);
    }
    for (int g = 0; g < n_dev; g++) {
        (void)dev_set(ids[g]);
        if (d[g]) (void)hipFree(d[g]);
        if (st[g]) (void)hipStreamDestroy(st[g]);
        if (comms[g]) (void)r.CommDestroy(comms[g]);
    }
    if (nrc != 0) return set_error(NH_EDEVICE, "ncclAllReduce: %s", r.GetErrorString(nrc));
    if (he != hipSuccess) return set_error(NH_EDEVICE, "allreduce_counters: %s", hipGetErrorString(he));
    char buf[512];
    snprintf(buf, sizeof buf, "RCCL %d.%d.%d (%s), ncclCommInitAll over %d device(s), ncclAllReduce(4 x uint64, sum)%s",
             version / 10000, (version / 100) % 100, version % 100, r.path.c_str(), n_dev,
             d_src ? " of the device-resident counters" : "");
    backend = buf;
    return NH_OK;
}

}  // namespace nh

extern "C" int nh_allreduce_counters(const int32_t *device_ids, int32_t n_devices, uint64_t *counters,
                                     char *backend, size_t backend_len) {
    std::string b;
    std::vector<int> ids;
    if (device_ids)
        ids.assign(device_ids, device_ids + (n_devices > 0 ? n_devices : 0));
    else
        for (int i = 0; i < n_devices; i++) ids.push_back(i);
    const int rc = nh::allreduce_counters(ids.data(), (int)ids.size(), counters, b, nullptr);
    if (backend && backend_len) snprintf(backend, backend_len, "%s", rc ? nh::g_last_error.c_str() : b.c_str());
    return rc;
}
