// LDS-DMA GEMM kernels, operand layout "tt" (A_KC = false, B_KC = true): see tf_gemm_dma.h.
#include "tf_gemm_dma_launch.h"
namespace tf {
template void launch_dma_plan<false, true>(int, const PlainOp&, const PlainOp&, const GemmEpi&, int, int, int, int, int, void*);
}
