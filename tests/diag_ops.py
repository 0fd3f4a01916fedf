"""Op-level parity at the failing full-size shapes (diagnostic)."""
import sys, os, traceback
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import kernel_cases as kc
def run(name, fn, *a):
    try:
        fn("cuda", *a); print("OK  ", name, a, flush=True)
    except Exception as e:
        print("FAIL", name, a, str(e)[:200], flush=True)
for a in [(2, 16, 16, 1512, 1512, 3, 2, 63), (2, 10, 44, 1512, 1512, 3, 2, 63), (2, 16, 16, 576, 576, 3, 1, 24), (2, 16, 16, 216, 216, 3, 2, 9), (2, 8, 8, 1512, 1512, 3, 1, 63)]:
    run("conv", kc.check_conv, *a)
for a in [(348, 6048, 1512), (348, 1512, 6048), (348, 1512, 1512), (348, 4536, 1512), (512, 1512, 576), (128, 1512, 1512), (348, 576, 2304)]:
    run("gemm", kc.check_gemm, *a)
run("attention", kc.check_attention, 2, 4, 174, 378)
run("attention", kc.check_attention, 2, 4, 174, 144)
run("ln", kc.check_layernorm, 348, 1512)
run("bn", kc.check_bn, 2, 8, 8, 1512, True, True)
run("bn", kc.check_bn, 2, 16, 16, 1512, True, False)
