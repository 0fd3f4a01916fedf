// tf_gemm_f32, operand layout A [k][m], B [k][n] (weight gradients; also the swapped small-Cout form): the register-staged engine instantiations + autotuner for this layout (one TU per layout).
#include "tf_gemm_engine.h"

namespace tf {
int gemm_plain_tn(const PlainOp& a, const PlainOp& b, const GemmEpi& ep, int M, int N, int K, int batch, bool allow_splitk, void* stream, const char* what) {
    return launch_gemm<PlainOp, false, PlainOp, false>(a, b, ep, M, N, K, batch, allow_splitk, stream, what);
}
}  // namespace tf
