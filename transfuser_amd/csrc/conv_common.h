// Shared by the implicit-GEMM convolution translation units (conv_fwd / conv_dgrad / conv_wgrad / conv_stem.cpp: one per direction so that
// hipcc compiles the engine instantiations of the four im2col loader families in parallel).
#pragma once
#include "tf_common.h"
#include "../../include/transfuser_hip.h"

namespace tf {
inline int check_geom(const tf_conv_geom* g, const char* who) {
    TF_REQUIRE(g, "%s: null geometry", who);
    TF_REQUIRE(g->groups >= 1 && g->Cin % g->groups == 0 && g->Cout % g->groups == 0, "%s: bad groups", who);
    TF_REQUIRE(g->ksize >= 1 && g->ksize <= 7 && g->stride >= 1 && g->pad >= 0, "%s: ksize %d / stride %d / pad %d unsupported (1 <= ksize <= 7)", who, g->ksize, g->stride, g->pad);
    TF_REQUIRE(g->Ho == (g->Hi + 2 * g->pad - g->ksize) / g->stride + 1 && g->Wo == (g->Wi + 2 * g->pad - g->ksize) / g->stride + 1,
               "%s: output size mismatch", who);
    return 0;
}

}  // namespace tf
