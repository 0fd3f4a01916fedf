#!/usr/bin/env python
"""Headline benchmark: TransFuser training samples/s (one RGB+LiDAR pair = one sample), bs=10/GPU.

  python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

A "step" = one full training iteration on a resident synthetic batch: zero grads ->
LidarCenterNet forward (RegNetY-3.2GF x2, 4 GPT stages x 4 layers, decoders, CenterNet head, GRU) ->
11 losses -> weighted sum -> backward -> [gradient all-reduce] -> AdamW, fp32, dropout p=0.1, every
kernel hand-written HIP.  Workload = BASELINE.json configs[1]: B=10, 3x256x704 RGB + 2x256x256 BEV
(+1x256x256 target-point channel), captured into one hipGraph.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import faulthandler

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_SAMPLE = {160: 230.3, 256: 266.6}  # algorithmic training FLOPs (3x forward), SURVEY.md section 8(d)
PEAK_F32_MFMA_TF = 157.3                     # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)


PMC_TRAFFIC_BYTES_PER_LAUNCH = int((2 * 274137.9 + 41107.5) * 1024)   # see dominant_kernel_roofline.__doc__


def dominant_kernel_roofline(dev, iters=20):
    """Dominant kernel = the fp32 MFMA GEMM engine (tf::gemm_kernel, ~70 % of the step's kernel time); its single most expensive
    call is GPT-4's fc1 [1740 x 1512] . [1512 x 6048] (+bias+ReLU).  Timed live with HIP events on the launch stream right after
    the sustained training loop; algorithmic FLOPs = 2*M*N*K per launch.  ``traffic``: fabric-side bytes per launch from the
    committed PMC run of the same kernel (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE doubled per the
    gfx950 note of MI355X_MICROARCH.md; profiles/r01_pmc_gemm_roofline.txt) - PMC counters cannot be read from inside this process."""
    from transfuser_amd import ops
    M, K, N = 1740, 1512, 6048
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.02
    b = torch.zeros(N, device=dev)
    out = torch.empty(M, N, device=dev)
    for _ in range(3):
        ops.linear_fwd(x, w, b, relu=True, out=out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        ops.linear_fwd(x, w, b, relu=True, out=out)
    e1.record()
    e1.synchronize()
    sec = e0.elapsed_time(e1) / 1e3 / iters
    flops = 2.0 * M * N * K
    ach = flops / sec / 1e12
    return dict(bound="mfma", kernel="tf::gemm_kernel<BM,BN,WM,BK,PlainOp,KC,PlainOp,KC,vec> (autotuned tiling) on GPT4 mlp.0: [1740x1512].[1512x6048], bias+ReLU epilogue",
                achieved=round(ach, 2), peak=PEAK_F32_MFMA_TF, unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TF, 4),
                flops_per_launch=flops, avg_launch_us=round(sec * 1e6, 2), traffic=PMC_TRAFFIC_BYTES_PER_LAUNCH,
                traffic_source="profiles/r01_pmc_gemm_roofline.txt (2*FETCH_SIZE + WRITE_SIZE, 64x64x16 tiling; L2 fabric requests incl. Infinity-Cache hits; "
                               "algorithmic bytes 89.2 MB)")


def cpu_baseline(cfg_factory, H, W):
    """The oracle (CPU restatement of the reference path, backbone pinned bit-exact to the reference's own
    transfuser.py) timed on this host: B=2 train steps of oracle.model_cpu.train_step, bounded to ~10-40 s."""
    from oracle import hist, model_cpu
    from transfuser_amd.data import synthetic_batch
    threads = min(os.cpu_count(), 64)   # oneDNN/OpenMP does not scale past this on these layer sizes
    torch.set_num_threads(threads)
    cfg = cfg_factory()
    torch.manual_seed(0)
    ref = model_cpu.LidarCenterNet(cfg, 'cpu', 'transFuser', use_velocity=False)
    ref.train()
    opt = model_cpu.make_optimizer(ref)
    B = 2
    batch = synthetic_batch(B, H, W, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192)
    t0 = time.time()
    model_cpu.train_step(ref, opt, batch, cfg)
    warm = time.time() - t0
    times = []
    while len(times) < 3 and sum(times) + warm < 30.0:
        t0 = time.time()
        model_cpu.train_step(ref, opt, batch, cfg)
        times.append(time.time() - t0)
    dt = sorted(times)[len(times) // 2] if times else warm
    return dict(value=round(B / dt, 3), unit="samples/s", cores=threads, kind="port",
                sample="oracle.model_cpu.train_step (PyTorch-CPU fp32, %d threads of %d cores), B=2, %dx%d, 1 warm-up + %d timed step(s), median" %
                       (threads, os.cpu_count(), H, W, len(times)))


T0 = time.perf_counter()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--watchdog", type=int, default=600, help="dump all Python stacks to stderr if still running after this many seconds")
    args = ap.parse_args()
    faulthandler.dump_traceback_later(args.watchdog, repeat=True, file=sys.stderr)
    torch.set_num_threads(min(16, os.cpu_count()))   # host-side glue only; the CPU baseline sets its own count

    def log(msg):
        print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)

    from transfuser_amd import _lib, ops
    from transfuser_amd.config import GlobalConfig
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd.train import Engine, init_distributed
    _lib.load()   # fails loudly when libtransfuser_hip.so is missing
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    rank, local_rank, world = init_distributed()
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    H, W, B = args.height, 704, args.batch

    def make_cfg():   # train.py defaults: n_layer 4, use_target_point_image 1, use_velocity 0, multitask, dropout .1
        cfg = GlobalConfig()
        cfg.n_layer = 4
        cfg.use_target_point_image = True
        cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = args.dropout
        return cfg

    cfg = make_cfg()
    torch.manual_seed(0)
    model = LidarCenterNet(cfg, dev, 'transFuser', 'regnety_032', 'regnety_032', use_velocity=False)
    model.train()
    hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).to(dev)[None])[0].cpu().numpy()
    batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, W, seed=rank, hist_fn=hist_fn).items()}
    log("model + batch on device")
    eng = Engine(model, cfg, lr=cfg.lr, use_graph=not args.no_graph)
    log("engine ready (arena %.1f M floats)" % (eng.arena.numel / 1e6))

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        eng.train_step(batch)
        if i == 0:
            torch.cuda.synchronize()
            log("first step done (incl. graph capture)" if not args.no_graph else "first eager step done")
    sync()
    log("warm-up done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tot, det = eng.train_step(batch)
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t)
    loss = float(tot)
    log("timed region done: %.2f ms/step" % (dt / args.steps * 1e3))
    assert loss == loss, "NaN loss"
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        gf = GFLOP_PER_SAMPLE.get(H)
        res = {
            "metric": "training samples/sec (RGB+LiDAR pair), bs=10/GPU", "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "TransFuser LidarCenterNet (RegNetY-3.2GF x2, 4 GPT x 4 layers, 168.0 M params) full train step, "
                                   "B=%d/GPU, 3x%dx%d RGB + 3x256x256 BEV, fp32, dropout %.2f, %s" %
                                   (B, H, W, args.dropout, "hipGraph replay" if not args.no_graph else "eager"),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(loss, 4)},
        }
        roof = dominant_kernel_roofline(dev)
        log("roofline microbench done")
        if gf:
            step_tf = value / world * gf / 1e3
            roof["step_achieved_tflops"] = round(step_tf, 2)
            roof["step_frac"] = round(step_tf / PEAK_F32_MFMA_TF, 4)
        res["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(make_cfg, H, W)
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
