// LDS-DMA GEMM kernels, operand layout "nn" (A_KC = true, B_KC = false): see tf_gemm_dma.h.
#include "tf_gemm_dma_launch.h"
namespace tf {
template void launch_dma_plan<true, false>(int, const PlainOp&, const PlainOp&, const GemmEpi&, int, int, int, int, int, void*);
}
