// Runtime layer of the C ABI: errors, handles, memory.  gfx950 / ROCm only.
#include "ddh_common.h"

#include <map>
#include <mutex>

namespace ddh {

static thread_local std::string g_error;
static std::mutex g_mutex;
static std::map<ddh_handle, HandleBase *> g_handles;
static ddh_handle g_next = 1;

void set_error(const std::string &msg) { g_error = msg; }
int fail(const std::string &msg) {
    g_error = msg;
    return -1;
}
int check_hip(hipError_t e, const char *what) {
    if (e == hipSuccess) return 0;
    g_error = std::string(what) + ": " + hipGetErrorString(e);
    return -2;
}
ddh_handle register_handle(HandleBase *h) {
    std::lock_guard<std::mutex> lock(g_mutex);
    ddh_handle id = g_next++;
    g_handles[id] = h;
    return id;
}
HandleBase *lookup_handle(ddh_handle h, HandleKind kind) {
    std::lock_guard<std::mutex> lock(g_mutex);
    auto it = g_handles.find(h);
    if (it == g_handles.end() || it->second->kind != kind) {
        g_error = "invalid handle";
        return nullptr;
    }
    return it->second;
}

static void *g_alias_buf = nullptr;
static size_t g_alias_cap = 0;

int resolve_alias(const double **in, const double *out, size_t in_elems, size_t out_elems, hipStream_t s) {
    const char *a0 = (const char *)*in, *a1 = a0 + in_elems * sizeof(double);
    const char *b0 = (const char *)out, *b1 = b0 + out_elems * sizeof(double);
    if (a1 <= b0 || b1 <= a0) return 0;
    const size_t bytes = in_elems * sizeof(double);
    if (bytes > g_alias_cap) {
        if (g_alias_buf) DDH_HIP(hipFree(g_alias_buf));       // hipFree waits for work that may still read it
        g_alias_buf = nullptr;
        g_alias_cap = 0;
        DDH_HIP(hipMalloc(&g_alias_buf, bytes + 256));
        g_alias_cap = bytes;
    }
    DDH_HIP(hipMemcpyAsync(g_alias_buf, *in, bytes, hipMemcpyDeviceToDevice, s));
    *in = (const double *)g_alias_buf;
    return 0;
}

}  // namespace ddh

using namespace ddh;

extern "C" {

const char *ddh_last_error(void) { return g_error.c_str(); }

int ddh_device_count(int *count) {
    DDH_HIP(hipGetDeviceCount(count));
    return 0;
}

int ddh_init(int device) {
    int n = 0;
    DDH_HIP(hipGetDeviceCount(&n));
    if (n <= 0) return fail("ddh_init: no HIP device visible");
    if (device < 0 || device >= n) return fail("ddh_init: device index out of range");
    DDH_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    DDH_HIP(hipGetDeviceProperties(&prop, device));
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos)
        return fail(std::string("ddh_init: built for gfx950, found ") + prop.gcnArchName);
    return 0;
}

int ddh_alloc(void **ptr, size_t bytes) {
    DDH_HIP(hipMalloc(ptr, bytes));
    return 0;
}
int ddh_free(void *ptr) {
    DDH_HIP(hipFree(ptr));
    return 0;
}
int ddh_memset(void *ptr, int value, size_t bytes, void *stream) {
    DDH_HIP(hipMemsetAsync(ptr, value, bytes, as_stream(stream)));
    return 0;
}
int ddh_memcpy_h2d(void *dst, const void *src_h, size_t bytes, void *stream) {
    DDH_HIP(hipMemcpyAsync(dst, src_h, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return 0;
}
int ddh_memcpy_d2h(void *dst_h, const void *src, size_t bytes, void *stream) {
    DDH_HIP(hipMemcpyAsync(dst_h, src, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return 0;
}
int ddh_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream) {
    DDH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, as_stream(stream)));
    return 0;
}
int ddh_stream_sync(void *stream) {
    DDH_HIP(hipStreamSynchronize(as_stream(stream)));
    return 0;
}
int ddh_destroy(ddh_handle h) {
    HandleBase *p = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mutex);
        auto it = g_handles.find(h);
        if (it == g_handles.end()) return fail("ddh_destroy: invalid handle");
        p = it->second;
        g_handles.erase(it);
    }
    delete p;
    return 0;
}

}  // extern "C"
