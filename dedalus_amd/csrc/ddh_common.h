// Shared internals of libdedalus_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/dedalus_hip.h"

namespace ddh {

void set_error(const std::string &msg);
int fail(const std::string &msg);                 // sets error, returns -1
int check_hip(hipError_t e, const char *what);    // 0 or negative

#define DDH_HIP(call)                                            \
    do {                                                         \
        int _s = ::ddh::check_hip((call), #call);                \
        if (_s) return _s;                                       \
    } while (0)

enum HandleKind : uint32_t { H_FFT = 1, H_MMT = 2, H_PENCIL = 3, H_GMMT = 4, H_STERMS = 5, H_CGEMV = 6, H_ELLT = 7,
                             H_COMM = 8, H_A2A = 9, H_DENSEINV = 10, H_ELLBAND = 11 };

struct HandleBase {
    HandleKind kind;
    virtual ~HandleBase() {}
};

ddh_handle register_handle(HandleBase *h);
HandleBase *lookup_handle(ddh_handle h, HandleKind kind);

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// In-place safety of the transform entry points (the reference hands forward(gdata, cdata, axis) two views of ONE
// field buffer, core/basis.py:185-193, core/transforms.py:43-51): when the input range overlaps the output range the
// input is first copied (stream-ordered) into a library-owned scratch arena and the kernel reads from there.
// On return *in points at the data to read.  Distinct buffers cost nothing.
int resolve_alias(const double **in, const double *out, size_t in_elems, size_t out_elems, hipStream_t s);

// Observed dispatch places block b on XCD b % 8; give each XCD a contiguous run of logical
// blocks so neighbouring tiles (which share 128-B lines) share an L2.  Speed only.
__device__ __forceinline__ unsigned xcd_swizzle(unsigned bid, unsigned nblocks) {
    const unsigned per = nblocks >> 3;
    if (per == 0 || bid >= (per << 3)) return bid;
    return (bid & 7u) * per + (bid >> 3);
}

}  // namespace ddh
