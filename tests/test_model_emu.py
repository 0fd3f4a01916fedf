"""End-to-end parity of the product model (HIP kernels, emulated on the host) against the CPU oracle on a
tiny RegNetY trunk: 11 losses, every parameter gradient, state_dict key identity, one optimizer step."""
import pytest
import torch

import model_cases as mc


@pytest.fixture(autouse=True)
def _backend(emu_backend):
    yield


def test_tiny_model_losses_and_grads():
    cfg = mc.tiny_config(n_layer=2)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu")
    batch = mc.small_batch(2, 32, 64, 64, 40)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr, verbose=True)


def test_tiny_model_with_velocity_and_odd_image():
    cfg = mc.tiny_config(n_layer=1)
    prod, ref = mc.build_pair(cfg, "regnety_tiny", "cpu", use_velocity=True)
    batch = mc.small_batch(1, 64, 96, 64, 40, seed=3)
    lp, lr = mc.run_pair(prod, ref, cfg, batch, "cpu")
    mc.compare(prod, ref, lp, lr)
