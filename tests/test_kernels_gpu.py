"""Op-level parity on a real MI355X: every HIP kernel (through the C ABI of libtransfuser_hip.so)
against a plain PyTorch fp32 reference / the CPU oracle.  Same cases as tests/test_kernels_emu.py."""
import pytest
import torch

import kernel_cases as kc


pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _gpu():
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    from transfuser_amd import _lib
    assert not _lib.is_test_backend()
    _lib.load()  # raises if libtransfuser_hip.so is missing: no fallback
    yield
    torch.cuda.synchronize()


@pytest.mark.parametrize("case", kc.GEMM_CASES, ids=str)
def test_gemm(case):
    kc.check_gemm("cuda", *case)


@pytest.mark.parametrize("case", kc.BATCHED_GEMM_CASES, ids=str)
def test_attention_gemms(case):
    kc.check_attention("cuda", *case)


@pytest.mark.parametrize("case", kc.CONV_CASES, ids=str)
def test_conv(case):
    kc.check_conv("cuda", *case)


@pytest.mark.parametrize("case", kc.FUSED_ATTENTION_CASES + kc.FUSED_ATTENTION_CASES_GPU, ids=str)
def test_fused_attention(case):
    kc.check_fused_attention("cuda", *case)


def test_stem_conv():
    kc.check_stem("cuda", 2, 12, 20)
    kc.check_stem("cuda", 3, 37, 51)      # odd sizes: ragged last pixel tile / partial panel of the direct kernels


@pytest.mark.parametrize("case", [(37, 72), (9, 216), (5, 1512), (6, 576), (3, 2048), (4, 70), (2, 2052), (130, 288)], ids=str)
def test_layernorm(case):
    kc.check_layernorm("cuda", *case)


def test_softmax():
    kc.check_softmax("cuda", 23, 174, 176)


@pytest.mark.parametrize("case", [(2, 6, 7, 72, True, True), (1, 5, 5, 216, False, False), (3, 4, 4, 32, True, False), (2, 3, 5, 7, False, True)], ids=str)
def test_batchnorm(case):
    kc.check_bn("cuda", *case)


def test_bn_statistics_fused_into_the_producing_conv():
    kc.check_bn_fused_stats("cuda")


def test_bn_eval():
    kc.check_bn_eval("cuda")


def test_reductions_with_finalize():
    """BatchNorm backward / column sums / SE gate gradient through the reduce + finalize launches (kernel_cases.check_fused_finalize: the checks
    written for the removed "last block finishes" form still pin the two-launch path)."""
    kc.check_fused_finalize("cuda")


def test_skinny_wgrad():
    kc.check_skinny_wgrad("cuda")


def test_colsum_multi():
    kc.check_colsum_multi("cuda")


def test_colsum_and_se():
    kc.check_colsum("cuda")
    kc.check_se("cuda", 3, 5, 6, 72)


@pytest.mark.parametrize("case", [(2, 40, 44, 24, 5, 22), (2, 16, 16, 8, 8, 8), (1, 13, 9, 4, 5, 4), (2, 5, 22, 12, 5, 22), (2, 3, 7, 8, 5, 22), (1, 1, 2, 4, 5, 22)], ids=str)
def test_pool_tokens(case):
    kc.check_pool_tokens("cuda", *case)


@pytest.mark.parametrize("case", kc.BILINEAR_CASES, ids=str)
def test_bilinear(case):
    kc.check_bilinear("cuda", *case)


def test_losses():
    kc.check_ce("cuda", 300, 7, False)
    kc.check_ce("cuda", 257, 3, True)
    kc.check_l1("cuda", 1000, True)
    kc.check_l1("cuda", 80, False)


def test_gru_and_misc():
    kc.check_gru("cuda", 5, 64)
    kc.check_misc("cuda")


def test_adamw():
    kc.check_adamw("cuda", 1003)


def test_lidar_hist():
    kc.check_hist("cuda", 2, 3000)
    kc.check_hist("cuda", 3, 3001, stride=5)


@pytest.mark.parametrize("N", [32768, 40000], ids=["bench_cloud_10x32768", "max_lidar_points_10x40000"])
def test_lidar_hist_full_size_ragged(N):
    """H1 at SURVEY 8d's bench cloud (10 x 32768) and at the reference's cap (max_lidar_points = 40000, config.py) with ragged num_points
    (a full, an empty, a one-point sample): bit-exact against the oracle's np.histogramdd restatement."""
    kc.check_hist("cuda", 10, N, ragged=[N, N - 777, N, 0, 1, N // 2, 255, 257, N - 1, 12345])


def test_lidar_camera_correspondences():
    kc.check_correspondences("cuda")


@pytest.mark.parametrize("case", [(3, False), (2, True)], ids=str)
def test_centernet_targets_and_losses(case):
    kc.check_centernet("cuda", *case)


@pytest.mark.parametrize("plan", kc.ENGINE_PLANS, ids=str)
def test_engine_tilings(plan):
    kc.check_engine_plan("cuda", *plan)


@pytest.mark.parametrize("cfg", kc.DMA_KINDS, ids=str)
def test_gemm_dma_configurations(cfg):
    kc.check_gemm_dma("cuda", cfg[0], cfg[1], kc.DMA_SHAPES_GPU)


@pytest.mark.parametrize("cfg", kc.BF16_PLANS, ids=str)
def test_bf16_mfma_mode(cfg):
    kc.check_bf16_mode("cuda", cfg[0], cfg[1])


@pytest.mark.parametrize("cfg", kc.BF16_PLANS, ids=str)
def test_f32x3_split_mode(cfg):
    kc.check_f32x3_mode("cuda", cfg[0], cfg[1])


@pytest.mark.parametrize("case", kc.GROUPED_CONV_CASES, ids=str)
def test_grouped_conv_direct(case):
    kc.check_conv_grouped("cuda", *case)


@pytest.mark.parametrize("case", kc.GROUPED_S2_CASES, ids=str)
def test_conv_grouped_stride2_direct_kernels(case):
    kc.check_conv_grouped_s2("cuda", *case)


def test_conv_grouped_stride2_direct_kernels_compute_modes():
    kc.check_grouped_s2_modes("cuda")


def test_im2col_gemm_form_of_few_row_deep_k_convolutions():
    kc.check_im2col_gemm_conv("cuda")


@pytest.mark.parametrize("cfg", kc.TWO_PASS_CASES, ids=str)
def test_two_pass_splitk(cfg):
    kc.check_two_pass_splitk("cuda", *cfg)


@pytest.mark.parametrize("kind", kc.STREAM_K_KINDS_GPU, ids=str)
def test_stream_k_plans(kind):
    kc.check_stream_k("cuda", kind)


def test_f32x3_split_mode_direct_convs():
    kc.check_f32x3_direct("cuda")


def test_bf16_mfma_mode_direct_convs():
    kc.check_bf16_direct("cuda")


@pytest.mark.parametrize("case", kc.GATHER_CASES, ids=str)
def test_gather_sum(case):
    kc.check_gather_sum("cuda", *case)


@pytest.mark.parametrize("case", kc.PILLAR_CASES, ids=str)
def test_point_pillars(case):
    kc.check_pillars("cuda", *case)


def test_pillar_index_three_launch_form_equals_the_seven_launch_form():
    kc.check_pillar_index_forms("cuda")


@pytest.mark.parametrize("case", kc.SE_EXCITE_CASES, ids=str)
def test_se_excite_fused(case):
    kc.check_se_excite("cuda", *case)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_storage_cast_and_gemm(mode):
    kc.check_lowp16_storage("cuda", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_copies_written_by_their_producers(mode):
    kc.check_lowp16_fused_producers("cuda", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_bottleneck_conv_operand_producers(mode):
    kc.check_lowp16_conv_producers("cuda", mode)


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
def test_lowp16_bottleneck_convs_on_stored_operands_stage(mode):
    kc.check_lowp16_conv_stage("cuda", mode)


def test_layernorm_backward_with_dropped_second_output():
    kc.check_layernorm_bwd_drop("cuda")


def test_fp16_mfma_mode():
    kc.check_bf16_mode("cuda", "plan", (64, 64, 16, 1), mode="fp16")
    kc.check_bf16_mode("cuda", "dma", (1, 1), mode="fp16")


def test_conv1x1_stride2_dgrad_gemm_scatter():
    kc.check_conv1x1_s2_dgrad("cuda")


@pytest.mark.parametrize("case", [(3, 6, 10, 48, 12), (2, 16, 44, 576, 144), (10, 5, 7, 72, 8)], ids=str)
def test_bn_apply_folded_into_se_consumers(case):
    kc.check_bn_se_consumer_fusion("cuda", *case)


def test_convnext_block_pieces():
    kc.check_convnext_pieces("cuda")


def test_resnet_stem_conv7x7_and_maxpool():
    kc.check_resnet_stem_and_pool("cuda")


@pytest.mark.parametrize("case", kc.THIN_CONV_CASES, ids=str)
def test_conv_thin_output(case):
    kc.check_conv_thin("cuda", *case)


@pytest.mark.parametrize("case", kc.DIRECT_CONV_CASES, ids=str)
def test_conv_direct_small_channels(case):
    kc.check_conv_direct("cuda", *case)


@pytest.mark.parametrize("case", [(2, 128, 352, 32, 32), (1, 256, 704, 32, 7), (3, 130, 197, 32, 1)], ids=str)
def test_conv_direct_full_resolution(case):
    """Decoder-tail shapes at (near) full resolution: these take the direct kernels through the normal size threshold."""
    B, H, W, Cin, Cout = case
    from transfuser_amd import ops
    assert ops._direct_ok((B, H, W, Cin), Cout, Cin, 3, 1, 1, 1)
    kc.check_conv("cuda", B, H, W, Cin, Cout, 3, 1, 1)


@pytest.mark.parametrize("case", kc.DECODE_CASES, ids=str)
def test_centernet_decode(case):
    kc.check_centernet_decode("cuda", *case)


@pytest.mark.parametrize("case", [(130, 72, 96), (10, 64, 32), (200, 50, 150)], ids=str)
def test_gemm_relu_mask_epilogue(case):
    kc.check_gemm_mask("cuda", *case)


@pytest.mark.parametrize("case", [(174, 72, 72), (348, 216, 864), (130, 100, 52)], ids=str)
def test_gemm_dropout_residual_epilogue(case):
    kc.check_gemm_dropout("cuda", *case)


@pytest.fixture
def tuned_plans():
    """The plan cache bench.py runs with (transfuser_amd/plans/mi355x.txt, loaded by train.Engine)."""
    import os
    from transfuser_amd import ops
    path = os.path.join(os.path.dirname(os.path.abspath(ops.__file__)), "plans", "mi355x.txt")
    assert ops.plans_load(path) > 100
    yield
    ops.L().tf_plans_clear()


@pytest.mark.parametrize("case", kc.BENCH_GEMMS, ids=str)
def test_bench_shape_gemms_with_tuned_plans(case, tuned_plans):
    kc.check_bench_gemm("cuda", *case)


@pytest.mark.parametrize("case", kc.BENCH_CONVS, ids=str)
def test_bench_shape_convs_with_tuned_plans(case, tuned_plans):
    kc.check_bench_conv("cuda", *case)


def test_dataprep_matches_reference_functions():
    """GPU-side batch preparation vs the reference's own data.py functions (tests/golden/dataprep.npz)."""
    import dataprep_cases as dc
    dc.check_dataprep("cuda")


@pytest.mark.parametrize("case", kc.GEMM_PAIR_CASES + [(7040, 576, 576, 6), (2560, 576, 576, 6), (28160, 216, 216, 64)], ids=str)
def test_gemm_pair_launch(case):
    kc.check_gemm_pair("cuda", *case, **({} if case[0] < 2000 else dict(bks=((32, 32), (32, 16)))))
