// Body of launch_dma_plan<A_KC, B_KC> (declared in tf_gemm_engine.h): kind -> kernel configuration of tf_gemm_dma.h.
// Included by gemm_dma_{nt,nn,tn,tt}.cpp, one operand layout per translation unit (keeps hipcc's per-file time down; built in parallel).
#pragma once
#include "tf_gemm_dma.h"

namespace tf {
template <bool A_KC, bool B_KC>
void launch_dma_plan(int kind, const PlainOp& la, const PlainOp& lb, const GemmEpi& ep, int M, int N, int K, int batch, int splitk, void* stream) {
    //                          TM TN WM WN BK ST              OCC
    switch (kind) {
    case 1: launch_dma_cfg<2, 2, 2, 2, 16, 3, A_KC, B_KC, 2>(la, lb, ep, M, N, K, batch, splitk, stream); break;   // 128 x 128
    case 2: launch_dma_cfg<1, 1, 2, 2, 16, 4, A_KC, B_KC, 4>(la, lb, ep, M, N, K, batch, splitk, stream); break;   //  64 x  64
    case 3: launch_dma_cfg<2, 1, 2, 2, 16, 3, A_KC, B_KC, 3>(la, lb, ep, M, N, K, batch, splitk, stream); break;   // 128 x  64
    case 4: launch_dma_cfg<1, 2, 2, 2, 16, 3, A_KC, B_KC, 3>(la, lb, ep, M, N, K, batch, splitk, stream); break;   //  64 x 128
    case 5: launch_dma_cfg<2, 2, 2, 2, 32, 2, A_KC, B_KC, 2>(la, lb, ep, M, N, K, batch, splitk, stream); break;   // 128 x 128, BK 32
    // 6-8: every wave owns a 64 x 64 accumulator block (4 fragments feed 4 MFMA tiles: the ratio the bf16x3 split wants) in SMALL workgroups
    case 6: launch_dma_cfg<2, 2, 1, 1, 16, 4, A_KC, B_KC, 2>(la, lb, ep, M, N, K, batch, splitk, stream); break;   //  64 x  64, one wave, no barrier
    case 7: launch_dma_cfg<2, 2, 1, 2, 16, 3, A_KC, B_KC, 2>(la, lb, ep, M, N, K, batch, splitk, stream); break;   //  64 x 128, two waves
    default: launch_dma_cfg<2, 2, 2, 1, 16, 3, A_KC, B_KC, 2>(la, lb, ep, M, N, K, batch, splitk, stream); break;  // 128 x  64, two waves
    }
}
}  // namespace tf
