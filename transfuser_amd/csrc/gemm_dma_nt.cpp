// LDS-DMA GEMM kernels, operand layout "nt" (A_KC = true, B_KC = true): see tf_gemm_dma.h.
#include "tf_gemm_dma_launch.h"
namespace tf {
template void launch_dma_plan<true, true>(int, const PlainOp&, const PlainOp&, const GemmEpi&, int, int, int, int, int, void*);
}
namespace tf {
bool dma_eligible(const PlainOp& a, const PlainOp& b) { return dma_eligible_impl(a, b); }
}
