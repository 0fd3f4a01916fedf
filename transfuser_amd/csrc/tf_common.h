// Host-side helpers shared by every C-ABI translation unit.
#pragma once
#include "tf_prims.h"
#include <stdio.h>
#include <string.h>

namespace tf {

void set_error(const char* fmt, ...);  // api.cpp
bool ablated(const char* kernel);      // api.cpp: TF_ABLATE=substr[,substr...] turns the matching kernels' launches into no-ops (timing diagnosis only)

#ifdef TF_EMU
#define TF_LAUNCH(kern, grid, block, stream, ...)                         \
    do {                                                                    \
        (void)(stream);                                                     \
        emu::launch((grid), (block), [=]() { kern(__VA_ARGS__); });         \
    } while (0)
inline int launch_status(const char*) { return 0; }
#else
#define TF_LAUNCH(kern, grid, block, stream, ...)                                                        \
    do {                                                                                                   \
        static const bool tf_ablated_ = tf::ablated(#kern);                                              \
        if (!tf_ablated_) hipLaunchKernelGGL(kern, (grid), (block), 0, (hipStream_t)(stream), __VA_ARGS__); \
    } while (0)
inline int launch_status(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}
#endif

#define TF_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            tf::set_error(__VA_ARGS__);  \
            return -1;                   \
        }                                \
    } while (0)

inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

}  // namespace tf
