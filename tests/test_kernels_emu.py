"""CPU checks of the HIP kernel sources, compiled for the host against the fiber emulator
(tests/emu).  Each op is compared with a plain PyTorch fp32 reference.  The same comparisons
run on the real GPU in tests/test_kernels_gpu.py (shared case lists in tests/kernel_cases.py)."""
import pytest
import torch

import kernel_cases as kc


@pytest.fixture(autouse=True)
def _backend(emu_backend):
    yield


@pytest.mark.parametrize("case", kc.GEMM_CASES, ids=str)
def test_gemm(case):
    kc.check_gemm("cpu", *case)


@pytest.mark.parametrize("case", kc.BATCHED_GEMM_CASES, ids=str)
def test_attention_gemms(case):
    kc.check_attention("cpu", *case)


@pytest.mark.parametrize("case", kc.CONV_CASES, ids=str)
def test_conv(case):
    kc.check_conv("cpu", *case)


@pytest.mark.parametrize("case", kc.FUSED_ATTENTION_CASES, ids=str)
def test_fused_attention(case):
    kc.check_fused_attention("cpu", *case)


def test_stem_conv():
    kc.check_stem("cpu", 2, 12, 20)
    kc.check_stem("cpu", 3, 37, 51)      # odd sizes: ragged last pixel tile / partial panel of the direct kernels


@pytest.mark.parametrize("case", [(37, 72), (9, 216), (5, 1512), (6, 576), (3, 2048), (4, 70), (2, 2052), (130, 288)], ids=str)
def test_layernorm(case):
    kc.check_layernorm("cpu", *case)


def test_softmax():
    kc.check_softmax("cpu", 23, 174, 176)


@pytest.mark.parametrize("case", [(2, 6, 7, 72, True, True), (1, 5, 5, 216, False, False), (3, 4, 4, 32, True, False), (2, 3, 5, 7, False, True)], ids=str)
def test_batchnorm(case):
    kc.check_bn("cpu", *case)


def test_bn_statistics_fused_into_the_producing_conv():
    kc.check_bn_fused_stats("cpu")


def test_bn_eval():
    kc.check_bn_eval("cpu")


def test_reductions_with_finalize():
    """BatchNorm backward / column sums / SE gate gradient through the reduce + finalize launches (kernel_cases.check_fused_finalize: the checks
    written for the removed "last block finishes" form still pin the two-launch path)."""
    kc.check_fused_finalize("cpu")


def test_skinny_wgrad():
    kc.check_skinny_wgrad("cpu")


def test_colsum_multi():
    kc.check_colsum_multi("cpu")


def test_colsum_and_se():
    kc.check_colsum("cpu")
    kc.check_se("cpu", 3, 5, 6, 72)


@pytest.mark.parametrize("case", [(2, 40, 44, 24, 5, 22), (2, 16, 16, 8, 8, 8), (1, 13, 9, 4, 5, 4), (2, 5, 22, 12, 5, 22), (2, 3, 7, 8, 5, 22), (1, 1, 2, 4, 5, 22)], ids=str)
def test_pool_tokens(case):
    kc.check_pool_tokens("cpu", *case)


@pytest.mark.parametrize("case", kc.BILINEAR_CASES, ids=str)
def test_bilinear(case):
    kc.check_bilinear("cpu", *case)


def test_losses():
    kc.check_ce("cpu", 300, 7, False)
    kc.check_ce("cpu", 257, 3, True)
    kc.check_l1("cpu", 1000, True)
    kc.check_l1("cpu", 80, False)


def test_gru_and_misc():
    kc.check_gru("cpu", 5, 64)
    kc.check_misc("cpu")


def test_adamw():
    kc.check_adamw("cpu", 1003)


def test_lidar_hist():
    kc.check_hist("cpu", 2, 3000)
    kc.check_hist("cpu", 3, 3001, stride=5)
    kc.check_hist("cpu", 4, 2049, ragged=[2049, 0, 1, 1500])
    kc.check_hist("cpu", 9, 1100, ragged=[1100, 0, 1, 700, 1100, 64, 65, 1024, 1025])       # more than 8 samples: the slab kernel's second round of sample -> XCD slots, > 1024 points: a second trip


def test_lidar_camera_correspondences():
    kc.check_correspondences("cpu")


@pytest.mark.parametrize("case", [(3, False), (2, True)], ids=str)
def test_centernet_targets_and_losses(case):
    kc.check_centernet("cpu", *case)


@pytest.mark.parametrize("plan", kc.ENGINE_PLANS, ids=str)
def test_engine_tilings(plan):
    kc.check_engine_plan("cpu", *plan)


@pytest.mark.parametrize("cfg", kc.DMA_KINDS, ids=str)
def test_gemm_dma_configurations(cfg):
    kc.check_gemm_dma("cpu", cfg[0], cfg[1], kc.DMA_SHAPES_SMALL)


@pytest.mark.parametrize("cfg", kc.BF16_PLANS, ids=str)
def test_bf16_mfma_mode(cfg):
    kc.check_bf16_mode("cpu", cfg[0], cfg[1])


@pytest.mark.parametrize("cfg", kc.BF16_PLANS, ids=str)
def test_f32x3_split_mode(cfg):
    kc.check_f32x3_mode("cpu", cfg[0], cfg[1])


@pytest.mark.parametrize("case", kc.GROUPED_CONV_CASES, ids=str)
def test_grouped_conv_direct(case):
    kc.check_conv_grouped("cpu", *case)


@pytest.mark.parametrize("case", kc.GROUPED_S2_CASES, ids=str)
def test_conv_grouped_stride2_direct_kernels(case):
    kc.check_conv_grouped_s2("cpu", *case)


@pytest.mark.parametrize("cfg", kc.TWO_PASS_CASES, ids=str)
def test_two_pass_splitk(cfg):
    kc.check_two_pass_splitk("cpu", *cfg)


@pytest.mark.parametrize("kind", kc.STREAM_K_KINDS_EMU, ids=str)
def test_stream_k_plans(kind):
    kc.check_stream_k("cpu", kind)


def test_f32x3_split_mode_direct_convs():
    kc.check_f32x3_direct("cpu")


def test_bf16_mfma_mode_direct_convs():
    kc.check_bf16_direct("cpu")


@pytest.mark.parametrize("case", kc.GATHER_CASES, ids=str)
def test_gather_sum(case):
    kc.check_gather_sum("cpu", *case)


@pytest.mark.parametrize("case", kc.PILLAR_CASES, ids=str)
def test_point_pillars(case):
    kc.check_pillars("cpu", *case)


def test_pillar_index_three_launch_form_equals_the_seven_launch_form():
    kc.check_pillar_index_forms("cpu")


@pytest.mark.parametrize("case", kc.SE_EXCITE_CASES, ids=str)
def test_se_excite_fused(case):
    kc.check_se_excite("cpu", *case)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_storage_cast_and_gemm(mode):
    kc.check_lowp16_storage("cpu", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_copies_written_by_their_producers(mode):
    kc.check_lowp16_fused_producers("cpu", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_bottleneck_conv_operand_producers(mode):
    kc.check_lowp16_conv_producers("cpu", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_bottleneck_convs_on_stored_operands_stage(mode):
    kc.check_lowp16_conv_stage("cpu", mode)


def test_layernorm_backward_with_dropped_second_output():
    kc.check_layernorm_bwd_drop("cpu")


def test_fp16_mfma_mode():
    kc.check_bf16_mode("cpu", "plan", (64, 64, 16, 1), mode="fp16")
    kc.check_bf16_mode("cpu", "dma", (1, 1), mode="fp16")


def test_conv1x1_stride2_dgrad_gemm_scatter():
    kc.check_conv1x1_s2_dgrad("cpu")


@pytest.mark.parametrize("case", [(3, 6, 10, 48, 12), (2, 16, 44, 576, 144), (10, 5, 7, 72, 8)], ids=str)
def test_bn_apply_folded_into_se_consumers(case):
    kc.check_bn_se_consumer_fusion("cpu", *case)


def test_convnext_block_pieces():
    kc.check_convnext_pieces("cpu")


def test_resnet_stem_conv7x7_and_maxpool():
    kc.check_resnet_stem_and_pool("cpu")


@pytest.mark.parametrize("case", kc.THIN_CONV_CASES, ids=str)
def test_conv_thin_output(case):
    kc.check_conv_thin("cpu", *case)


@pytest.mark.parametrize("case", kc.DIRECT_CONV_CASES, ids=str)
def test_conv_direct_small_channels(case):
    kc.check_conv_direct("cpu", *case)


@pytest.mark.parametrize("case", kc.DECODE_CASES, ids=str)
def test_centernet_decode(case):
    kc.check_centernet_decode("cpu", *case)


@pytest.mark.parametrize("case", [(130, 72, 96), (10, 64, 32), (200, 50, 150)], ids=str)
def test_gemm_relu_mask_epilogue(case):
    kc.check_gemm_mask("cpu", *case)


@pytest.mark.parametrize("case", [(174, 72, 72), (348, 216, 864), (130, 100, 52)], ids=str)
def test_gemm_dropout_residual_epilogue(case):
    kc.check_gemm_dropout("cpu", *case)


def test_conv_grouped_stride2_direct_kernels_compute_modes():
    kc.check_grouped_s2_modes("cpu")


def test_im2col_gemm_form_of_few_row_deep_k_convolutions():
    kc.check_im2col_gemm_conv("cpu")



@pytest.mark.parametrize("case", kc.GEMM_PAIR_CASES, ids=str)
def test_gemm_pair_launch(case):
    kc.check_gemm_pair("cpu", *case)
