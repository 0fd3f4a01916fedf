#!/usr/bin/env python
"""Headline benchmark: TransFuser training samples/s (one RGB+LiDAR pair = one sample), bs=10/GPU.

  python bench.py --gpus N --steps K --warmup W

N > 1 works both ways: launched by ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` (one rank per GPU, the
driver's command) or from a bare shell - without a torchrun environment bench.py spawns its N ranks itself (127.0.0.1 rendezvous),
relays rank 0's JSON line and exits with the worst rank's status.

A "step" = one full training iteration on a resident synthetic batch: zero grads -> LidarCenterNet forward
(RegNetY-3.2GF x2, 4 GPT stages x 4 layers, decoders, CenterNet head, GRU) -> 11 losses -> weighted sum -> backward -> [gradient
all-reduce over RCCL, overlapped with the backward of the earlier stages] -> AdamW, dropout p=0.1, every kernel hand-written HIP.
Default workload = BASELINE.json configs[1]: B=10, 3x256x704 RGB + 2x256x256 BEV (+1x256x256 target-point channel), fp32, hipGraph
replay.  ``--backbone geometric_fusion --batch 12 --height 160`` / ``--backbone latentTF --batch 16`` are configs[3] / [4].
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import faulthandler

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_SAMPLE = {("transFuser", 160): 230.3, ("transFuser", 256): 266.6, ("latentTF", 160): 230.3, ("latentTF", 256): 266.6,
                    ("geometric_fusion", 160): 110.2}   # algorithmic training FLOPs (3x forward), SURVEY.md section 8(d)
PEAK_F32_MFMA_TF = 157.3                                # /opt/skills/guides/MI355X_MICROARCH.md (v_mfma_f32_32x32x2_f32)
PEAK_BF16_MFMA_TF = 2500.0                              # dense bf16 MFMA peak of the same guide (never the 2:1-sparsity figure)
def _pmc_file(suffix=""):
    """Newest committed rocprofv3 PMC summary of the dominant GEMM (tools/pmc_roofline.sh -> profiles/rNN_pmc_gemm_roofline[_<precision>].json)."""
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        f = os.path.join(ROOT, "profiles", "%s_pmc_gemm_roofline%s.json" % (rnd, suffix))
        if os.path.exists(f):
            return f
    return ""


def _trace_file(suffix="", build_id=None):
    """Newest committed per-kernel summary of a rocprofv3 kernel trace of the graph-replayed bench step (tools/trace_csv_stats.py) that was
    measured ON THIS BUILD: the summary's first line names the library it traced ("# build_id <tf_build_id>", written by tools/gpu_round4.sh
    trace); a summary of another build - or one without that line - is not used (the figure would be stale against this run's FLOPs)."""
    for name in ("r06_kernel_trace_graph%s.txt", "r05_kernel_trace_graph%s.txt", "r04_kernel_trace_graph%s.txt"):
        f = os.path.join(ROOT, "profiles", name % suffix)
        if os.path.exists(f):
            try:
                first = open(f).readline().split()
            except OSError:
                continue
            if build_id and len(first) == 3 and first[:2] == ["#", "build_id"] and first[2] == build_id:
                return f
    return ""


DOMINANT = ("gemm a0b0", (1740, 6048, 1512, 1))         # GPT-4 mlp.0 forward: [1740 x 1512] . [1512 x 6048], bias + ReLU epilogue


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 10 transFuser, 12 geometric_fusion, 16 latentTF)")
    ap.add_argument("--height", type=int, default=None, help="RGB height (default 256; geometric_fusion only runs at 160)")
    ap.add_argument("--backbone", default="transFuser", choices=["transFuser", "geometric_fusion", "latentTF"])
    ap.add_argument("--dtype", default="f32", choices=["f32", "f32x3", "bf16", "fp16"],
                    help="f32 = exact fp32 MFMA, the reference's arithmetic (headline); f32x3 = the same fp32 data with every contraction as an exact "
                         "3-way bf16 split on the bf16 MFMA (six partial products, fp32-accurate; priced against the bf16 peak / 6); "
                         "bf16 = bf16-MFMA contractions with fp32 accumulate/storage/master weights (BASELINE configs[2])")
    ap.add_argument("--check", action="store_true",
                    help="after the timed region: one more step with the optimizer's wait for the all-reduce stream event-timed (per-rank allreduce_exposed_ms), "
                         "then all-gather a checksum of the parameter arena and ASSERT that every replica holds the same parameters (self-diagnosing --gpus N run)")
    ap.add_argument("--no-check", action="store_true", help="--gpus N > 1 runs the replica check by default (reported in the JSON line, fatal only with an explicit --check); this switches it off")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"], help="payload of the gradient all-reduce buckets (bf16 halves the xGMI bytes; the arena stays fp32)")
    ap.add_argument("--force-pieces", type=int, default=0,
                    help="N = 1 only: run the MULTI-GPU code path on this one GPU - a world-size-1 RCCL group, the backward cut into this many hipGraph pieces "
                         "(<= 8 = Engine.DEFAULT_CUTS + 1), every segment's gradient range all-reduced bucket by bucket on the side stream between the replays, the "
                         "reducer told it has 2 ranks (so the 1 / world scale runs), AdamW waiting for the side stream: what the per-GPU step costs when it is not ONE graph")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra measurements the default f32 line embeds (N=1 only): f32x3, BASELINE configs[2..4], the multi-GPU path on one GPU")
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--emulate-cpu", action="store_true",
                    help="DRY RUN of the multi-rank plumbing on a machine without GPUs (tests/test_distributed_cpu.py): tiny trunks, the HIP kernels "
                         "host-emulated (tests/emu), gloo - exercises the self-spawn, the cut / overlapped reducer path, --check and the no-teardown "
                         "exit; its numbers mean nothing and the line says so")
    ap.add_argument("--cpu-probe", type=int, default=0, help=argparse.SUPPRESS)      # child of cpu_baseline: B = 2 oracle steps on this many threads, prints the median seconds
    ap.add_argument("--watchdog", type=int, default=900, help="dump all Python stacks to stderr if still running after this many seconds")
    return ap.parse_args()


def spawn_ranks(args):
    """No torchrun environment and --gpus N > 1: be the launcher (one process per GPU, env:// rendezvous on 127.0.0.1)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    sys.exit(max(abs(rc) for rc in rcs))


def _is_engine_kind(kind):
    """Census kinds (ops._census_end) whose launches are MFMA-ENGINE kernels (gemm_kernel / gemm_pair_kernel / gemm_dma_kernel): every "gemm ..." row and
    the un-suffixed convolution rows; "conv fwd g", "conv dgrad g2", "conv wgrad*", "conv fwd t", ... are the direct kernels' rows."""
    return kind.startswith("gemm") or kind in ("conv fwd", "conv dgrad", "conv wgrad")


def dominant_kernel_roofline(eng, batch, dev, log, peak=PEAK_F32_MFMA_TF, headline_workload=False):
    """Roofline of the dominant kernel family (the fp32 MFMA GEMM engine, ~70 % of the step's kernel time) from INSIDE the step: one more
    training iteration is run eagerly with a HIP-event pair around every engine call (ops.census; events on the launch stream), and the
    figures are averages over that step's own launches - the same kernels, plans, operands and neighbours as in the timed region.
      achieved = 2 M N K of GPT-4's mlp.0 forward GEMM / its average in-step duration (4 launches per step);
      engine_* = all plain-GEMM + conv launches of the step (algorithmic FLOPs / event time);
      traffic  = fabric-side bytes per launch of the same kernel + plan from the committed rocprofv3 PMC passes (tools/pmc_roofline.sh ->
                 profiles/r02_pmc_gemm_roofline.json: 2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction of MI355X_MICROARCH.md); PMC
                 counters cannot be read from inside this process, so the number is read from that file at run time (null if absent)."""
    import torch
    from transfuser_amd import ops
    torch.cuda.synchronize()
    ops.census = []
    eng._eager_step(batch)
    torch.cuda.synchronize()
    rows, ops.census = ops.census, None
    dom = [(a.elapsed_time(b) * 1e-3, fl) for kind, shape, fl, a, b in rows if (kind, tuple(shape)) == DOMINANT]
    dom16 = not dom
    if dom16:   # 16-bit storage modes: the same product runs as tf_gemm16_nt_f32 (mlp.0 forward and mlp.2's input gradient share the shape)
        dom = [(a.elapsed_time(b) * 1e-3, fl) for kind, shape, fl, a, b in rows if kind == "gemm16 nt" and tuple(shape) == DOMINANT[1]]
    tot_s = sum(a.elapsed_time(b) for _, _, _, a, b in rows) * 1e-3
    tot_fl = sum(fl for _, _, fl, _, _ in rows)
    PEAK = peak
    roof = dict(bound="mfma", peak=PEAK, unit="TFLOP/s")
    # every (call kind, shape) of the step, ranked by the time it takes: the DOMINANT entry is the one with the most step time (round-4 verdict:
    # not the best-running shape), the five largest are listed; "gemm pair" = a layer's weight + input gradient in one grid (csrc/gemm_pair.cpp)
    groups = {}
    for kind, shape, fl, a, b in rows:
        g = groups.setdefault((kind, tuple(shape)), [0, 0.0, 0.0])
        g[0] += 1; g[1] += a.elapsed_time(b) * 1e-3; g[2] += fl
    ranked = sorted(groups.items(), key=lambda kv: -kv[1][1])
    fmt = lambda k, v: {"call": "%s %s" % (k[0], list(k[1])), "launches_per_step": v[0], "ms_per_step": round(v[1] * 1e3, 3), "gflop_per_launch": round(v[2] / v[0] / 1e9, 3),
                        "avg_launch_us": round(v[1] / v[0] * 1e6, 2), "tflops": round(v[2] / v[1] / 1e12, 2), "frac": round(v[2] / v[1] / 1e12 / PEAK, 4)}
    roof["top5"] = [fmt(k, v) for k, v in ranked[:5]]
    gpt = None
    if dom:
        sec = sum(t for t, _ in dom) / len(dom)
        ach = dom[0][1] / sec / 1e12
        gpt = dict(kernel=("tf::gemm_dma_kernel on 16-bit STORED operands (tf_gemm16_nt_f32) on GPT4 [1740x1512].[1512x6048] (mlp.0 forward, mlp.2 input gradient); "
                            if dom16 else "tf::gemm_kernel / tf::gemm_dma_kernel (autotuned plan) on GPT4 mlp.0 forward: [1740x1512].[1512x6048], bias+ReLU epilogue; ") +
                           "average of its %d launches inside one eager training step" % len(dom),
                    achieved=round(ach, 2), frac=round(ach / PEAK, 4), flops_per_launch=dom[0][1], avg_launch_us=round(sec * 1e6, 2))
        roof["gpt4_mlp0"] = gpt      # the best-running large GEMM of the step (the row earlier rounds quoted as THE roofline kernel): kept as one row
    if ranked:
        k, v = ranked[0]
        roof.update(kernel="%s %s (m, n, k, batch | B, Hi, Wi, Cin, Cout, ks, stride, groups): the MFMA-engine call with the most time in the step - %d launches, "
                           "%.2f ms of the %.2f ms the engine takes in one eager training step; HIP events around each launch on its stream" %
                           (k[0], list(k[1]), v[0], v[1] * 1e3, tot_s * 1e3),
                    achieved=round(v[2] / v[1] / 1e12, 2), frac=round(v[2] / v[1] / 1e12 / PEAK, 4), flops_per_launch=v[2] / v[0], avg_launch_us=round(v[1] / v[0] * 1e6, 2))
    roof.update(engine_calls=len(rows), engine_ms_per_step=round(tot_s * 1e3, 2), engine_tflops=round(tot_fl / tot_s / 1e12, 2),
                engine_frac=round(tot_fl / tot_s / 1e12 / PEAK, 4))
    # The event-timed figure above is an EAGER step: every one of ~700 engine launches carries its host launch gap.  The timed region replays
    # hipGraphs; the engine kernels' durations there come from the committed rocprofv3 kernel trace of this command (tools/gpu_round4.sh trace).
    try:
        bid = ops.L().tf_build_id().decode()
    except Exception:
        bid = None
    tr = _trace_file("" if peak == PEAK_F32_MFMA_TF else "_" + ops.get_precision(), bid)
    if not tr and headline_workload:
        roof["engine_graph"] = None      # no kernel-trace summary of THIS build is committed (profiles/r05_kernel_trace_graph*.txt names the build it traced)
    if tr and dom and headline_workload:      # the committed trace is of configs[1] (transFuser, B = 10, 256 x 704)
        try:
            import re
            txt = open(tr).read()
            ms = float(re.search(r"gemm engine[^,]*? ([0-9.]+)", txt).group(1))
            md = re.search(r"direct conv[^,]*? ([0-9.]+)", txt)
            # numerator and denominator cover the SAME launches (round-5 review, weak #3): the trace family "gemm engine" holds gemm_kernel / gemm_pair_kernel /
            # gemm_dma_kernel, i.e. the census kinds "gemm ..." and the un-suffixed "conv fwd / dgrad / wgrad" (implicit-GEMM engine convolutions); the
            # suffixed kinds (" g" / " g2" grouped, "*" decoder-tail, " t" thin-output) run on the direct kernels of the trace family "direct conv"
            eng_fl = sum(fl for kind, _, fl, _, _ in rows if _is_engine_kind(kind))
            roof["engine_graph"] = {"ms_per_step": ms, "gflop_per_step": round(eng_fl / 1e9, 1), "tflops": round(eng_fl / ms / 1e9, 2), "frac": round(eng_fl / ms / 1e9 / PEAK, 4),
                                    "source": "profiles/%s (traced on this build, %s): sum of the gemm_kernel / gemm_pair_kernel / gemm_dma_kernel durations per step in the rocprofv3 "
                                              "--kernel-trace of the hipGraph replay (B = 10, 256 x 704); FLOPs = this run's census rows that run on those kernels (kinds 'gemm *' and "
                                              "un-suffixed 'conv *'), NOT the direct / grouped / thin convolutions" % (os.path.basename(tr), bid)}
            if md:
                dms = float(md.group(1))
                roof["direct_conv_graph"] = {"ms_per_step": dms, "gflop_per_step": round((tot_fl - eng_fl) / 1e9, 1), "tflops": round((tot_fl - eng_fl) / dms / 1e9, 2),
                                             "frac": round((tot_fl - eng_fl) / dms / 1e9 / PEAK, 4),
                                             "source": "same trace: conv3x3_grouped / conv3x3_small / conv3x3_thin kernels; FLOPs = the census kinds with a ' g' / ' g2' / '*' / ' t' suffix"}
                roof["contractions_graph"] = {"ms_per_step": round(ms + dms, 2), "tflops": round(tot_fl / (ms + dms) / 1e9, 2), "frac": round(tot_fl / (ms + dms) / 1e9 / PEAK, 4)}
        except Exception as e:
            roof["engine_graph"] = {"error": "unreadable %s: %s" % (tr, e)}
    roof["traffic"], roof["traffic_source"] = None, "no PMC summary committed"
    pmc_file = _pmc_file("" if peak == PEAK_F32_MFMA_TF else "_f32x3" if peak < PEAK_BF16_MFMA_TF else "_" + ops.get_precision())
    if pmc_file and os.path.exists(pmc_file) and gpt is not None:
        try:
            pmc = json.load(open(pmc_file))
            gpt["traffic"] = int(pmc["traffic_bytes_per_launch"])
            gpt["traffic_source"] = "profiles/%s: %s" % (os.path.basename(pmc_file), pmc.get("note", ""))
            if pmc.get("mfma_busy_frac") is not None:
                gpt["mfma_busy_frac_pmc"] = pmc["mfma_busy_frac"]
        except Exception as e:   # a malformed summary must not kill the bench line
            gpt["traffic_source"] = "unreadable %s: %s" % (pmc_file, e)
    # counter traffic of the DOMINANT call: the trunk table (tools/pmc_trunk.sh -> profiles/r05_pmc_trunk.json) when it holds that shape
    trunk = next((f for f in (os.path.join(ROOT, "profiles", "%s_pmc_trunk.json" % r) for r in ("r06", "r05")) if os.path.exists(f)), "")
    if ranked and trunk and peak == PEAK_F32_MFMA_TF:
        try:
            k = ranked[0][0]
            want = {"gemm a0b0": "nt", "gemm a0b1": "nn", "gemm a1b1": "tn", "gemm pair wgrad+dgrad (one grid)": "pair"}.get(k[0])
            # census shapes are (m, n, k) of the call: forward (M, N, K); input gradient (M, K_in, N_out); weight gradient (N_out, K_in, M); pair = its input gradient's
            m, n, kk = (k[1][2], k[1][0], k[1][1]) if want == "tn" else (k[1][0], k[1][2], k[1][1]) if want in ("nn", "pair") else (k[1][0], k[1][1], k[1][2])
            for row in json.load(open(trunk)):
                if want and row["case"] == "gemm %s (%d,%d,%d)" % (want, m, n, kk) and row.get("traffic_bytes"):
                    roof["traffic"] = int(row["traffic_bytes"])
                    roof["traffic_source"] = ("profiles/" + os.path.basename(trunk) + ": 2 x FETCH_SIZE + WRITE_SIZE per call (rocprofv3 --pmc, separate passes, gfx950 correction); "
                                              "algorithmic %.1f MB; MFMA busy %.2f" % (row["algorithmic_bytes"] / 1e6, row.get("mfma_busy_frac") or -1))
        except Exception as e:
            roof["traffic_source"] = "unreadable %s: %s" % (trunk, e)
    log("in-step roofline census done (%d engine launches)" % len(rows))
    return roof


PEAK_HBM_TBS, ACHIEVABLE_HBM_TBS = 8.0, 6.3            # MI355X_MICROARCH.md: HBM3E spec / the sustained copy bandwidth the guide quotes


def hbm_rooflines(eng, batch, dev, log):
    """north_star's HBM clause, from INSIDE the step like the MFMA entry: one more eager training iteration with a HIP-event pair around every
    call of the bandwidth-bound kernels it names (ops.hbm_census: AdamW over the flat arena, BatchNorm forward / backward, LayerNorm
    forward; the attention softmax lives inside the fused attention kernel and never touches HBM), plus the H1 LiDAR histogram on the bench
    cloud (10 x 32768 points; not part of the timed step: batch preparation).  Per family: algorithmic bytes, event time,
    TB/s, fraction of the 8.0 TB/s spec and of the 6.3 TB/s achievable.  Eager launches carry their host launch gaps for the small layers
    (the late-stage BatchNorm / LayerNorm calls are launch-latency sized): AdamW and the stem-resolution BatchNorm are the bandwidth figures."""
    import torch
    from transfuser_amd import ops
    from transfuser_amd.data import synthetic_cloud
    torch.cuda.synchronize()
    ops.hbm_census = []
    eng._eager_step(batch)
    torch.cuda.synchronize()
    rows, ops.hbm_census = ops.hbm_census, None
    fam = {}
    for name, nbytes, a, b in rows:
        f = fam.setdefault(name, [0, 0, 0.0, 0.0, None])
        us = a.elapsed_time(b) * 1e3
        f[0] += 1; f[1] += nbytes; f[2] += us
        if f[4] is None or nbytes > f[4][0]:
            f[4] = (nbytes, us)
    pts = torch.from_numpy(synthetic_cloud(10, 32768, 0)).to(dev)
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):      # a replayed hipGraph of 20 calls, like the step itself (eager launches add ~4 us of host gap to each kernel)
        for _ in range(3):           # (on the capture stream: the counter workspace is per stream and must exist before the capture)
            ops.lidar_hist(pts)
        st.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(20):
                ops.lidar_hist(pts)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st); g.replay(); e1.record(st); e1.synchronize()
    h1_bytes = 10 * (32768 * 16 + 2 * 256 * 256 * 4)
    fam["H1 lidar histogram, 10 x 32768 points (16 B / point read + 512 KB / sample written, each once: one launch, counters in LDS - lidar_hist_slab_kernel; hipGraph replay of 20 calls)"] = \
        [20, 20 * h1_bytes, e0.elapsed_time(e1) * 1e3, 0.0, (h1_bytes, e0.elapsed_time(e1) * 1e3 / 20)]
    out = []
    for name, (calls, nbytes, us, _, big) in fam.items():
        tbs = nbytes / us / 1e6
        btbs = big[0] / big[1] / 1e6
        out.append({"kernel": name, "calls": calls, "bytes": int(nbytes), "us": round(us, 1), "achieved_TBps": round(tbs, 3), "frac_of_8.0": round(tbs / PEAK_HBM_TBS, 4),
                    "frac_of_6.3": round(tbs / ACHIEVABLE_HBM_TBS, 4),
                    "largest_call": {"bytes": int(big[0]), "us": round(big[1], 1), "achieved_TBps": round(btbs, 3), "frac_of_8.0": round(btbs / PEAK_HBM_TBS, 4)}})
    log("in-step HBM census done (%d calls)" % len(rows))
    res = {"bound": "hbm", "peak": PEAK_HBM_TBS, "achievable": ACHIEVABLE_HBM_TBS, "unit": "TB/s",
           "timing": "HIP events around each call inside one EAGER training step: the small calls (LayerNorm, late-stage BatchNorm) carry their host launch gaps, so their "
                     "TB/s is a launch-gap figure - 'pmc' below holds the counter-based table of the same kernels (kernel time + fabric bytes per call)", "kernels": out}
    # the counter-based table (tools/pmc_hbm.sh -> profiles/rNN_pmc_hbm.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, gfx950 correction, per call):
    # kernel-only time (no launch gaps), fabric-side traffic against the algorithmic bytes
    pm = next((f for f in (os.path.join(ROOT, "profiles", "%s_pmc_hbm.json" % r) for r in ("r06", "r05", "r04", "r03")) if os.path.exists(f)), "")
    if pm:
        try:
            res["pmc"] = {"source": "profiles/" + os.path.basename(pm),
                          "kernels": [{"case": r["case"], "algorithmic_bytes": int(r["algorithmic_bytes"]), "traffic_over_algorithmic": round(r["traffic_over_algorithmic"], 3),
                                       "kernel_us_per_call": round(r["kernel_us_per_call"], 2), "achieved_TBps": round(r["achieved_tbps_algorithmic"], 3),
                                       "frac_of_8.0": round(r["achieved_tbps_algorithmic"] / PEAK_HBM_TBS, 4), "launches_per_call": r.get("launches_per_call")}
                                      for r in json.load(open(pm))]}
        except Exception as e:
            res["pmc"] = {"error": "unreadable %s: %s" % (pm, e)}
    return res


def replica_check(eng, batch, rank, world, dev, log, cuda=True):
    """--check: (1) one more training step with the optimizer's wait for the all-reduce side stream bracketed by timing events - the time the
    main stream stalls there is the part of the gradient all-reduce the backward did NOT hide; (2) a checksum of the parameter arena (fp64
    sum + exact int64 sum of the raw bits) all-gathered over the ranks: data parallelism (train.py:134) keeps the replicas bit-identical, so
    any difference is a reduction bug and the run FAILS.  Every rank prints its own line to stderr; rank 0 puts the summary into the JSON."""
    import torch
    import torch.distributed as dist
    eng.reducer.time_waits = cuda
    eng.train_step(batch)
    if cuda:
        torch.cuda.synchronize()
    exposed = eng.reducer.exposed_ms() if cuda else 0.0
    eng.reducer.time_waits = False
    p = eng.arena.params[:eng.arena.active_numel]
    # the exact checksum is a 64-bit integer: it travels as two 32-bit halves (a float64 holds only 53 bits of the ~1e17 sum)
    bits = p.view(torch.int32).to(torch.int64).sum()
    mine = torch.stack([p.double().sum(), (bits >> 32).double(), (bits & 0xffffffff).double(), torch.tensor(exposed, dtype=torch.float64, device=dev)])
    rows = [torch.zeros_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(rows, mine)
    else:
        rows = [mine]
    rows = [r.cpu().tolist() for r in rows]
    rows = [[r[0], (int(r[1]) << 32) | int(r[2]), r[3]] for r in rows]
    print("[bench check] rank %d: allreduce_exposed_ms %.3f, param checksum %.9e / bits %d" % (rank, exposed, rows[rank][0], int(rows[rank][1])), file=sys.stderr, flush=True)
    equal = all(r[0] == rows[0][0] and r[1] == rows[0][1] for r in rows)
    assert equal, "replicas diverged: per-rank parameter checksums %s" % [(r[0], int(r[1])) for r in rows]
    log("replica check: %d rank(s) bit-identical; exposed all-reduce ms per rank %s" % (world, [round(r[2], 3) for r in rows]))
    return {"replicas_equal": True, "n_ranks": world, "allreduce_exposed_ms": [round(r[2], 3) for r in rows], "param_checksum": rows[0][0],
            "backward_pieces": eng.n_pieces(), "grad_dtype": "bf16" if eng.reducer.bf16 else "fp32"}


def cpu_probe(args):
    """Child process of cpu_baseline: the oracle's B = 2 training step on ``--cpu-probe`` threads (1 warm-up + 3 timed), one line 'CPU_PROBE <median seconds>'.
    A separate process because a PyTorch-CPU step cannot be interrupted from inside: the parent kills it at its deadline."""
    import torch
    from oracle import hist, model_cpu
    from transfuser_amd.config import GlobalConfig
    from transfuser_amd.data import synthetic_batch
    torch.set_num_threads(args.cpu_probe)
    cfg = GlobalConfig()
    cfg.n_layer = 4
    cfg.use_target_point_image = True
    cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = args.dropout
    torch.manual_seed(0)
    ref = model_cpu.LidarCenterNet(cfg, 'cpu', args.backbone, use_velocity=False)
    ref.train()
    opt = model_cpu.make_optimizer(ref)
    H = args.height or (160 if args.backbone == "geometric_fusion" else 256)
    keys = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic") + \
        (("bev_points", "cam_points") if args.backbone == "geometric_fusion" else ())
    b2 = {k: v for k, v in synthetic_batch(2, H, 704, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192).items() if k in keys}
    ts = []
    for i in range(4):
        t0 = time.time()
        model_cpu.train_step(ref, opt, b2, cfg)
        if i:
            ts.append(time.time() - t0)
    print("CPU_PROBE %.4f" % sorted(ts)[1], flush=True)


def cpu_baseline(make_cfg, backbone, H, W, budget_s=135.0, all_cores_deadline_s=75.0):
    """The oracle (CPU restatement of the reference path; its backbone is pinned bit-exact to the reference's own transfuser.py and
    its heads/losses to the reference's model.py) timed on this host with PyTorch-CPU fp32: B=2 (BASELINE configs[0], the reference's
    CPU-runnable case) with 2 warm-up + 5 timed steps, then B=10 (the workload of the GPU line) with 1 warm-up + up to 5 timed steps
    inside the time budget (SURVEY 8d asks for >= 5).  Thread count: min(64, cores) - measured on the 256-core GPU host: beyond 64 threads the oneDNN / OpenMP
    kernels of these layer sizes slow down (a 256-thread probe step did not finish in 15 minutes: torch.optim's per-tensor loop and the
    small convolutions thrash), so "all cores" is NOT the fastest configuration of the reference's CPU path; both numbers are stated."""
    import torch
    from oracle import hist, model_cpu
    from transfuser_amd.data import synthetic_batch
    ncpu = os.cpu_count()
    cfg = make_cfg()
    torch.manual_seed(0)
    ref = model_cpu.LidarCenterNet(cfg, 'cpu', backbone, use_velocity=False)
    ref.train()
    opt = model_cpu.make_optimizer(ref)
    keys = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic") + \
        (("bev_points", "cam_points") if backbone == "geometric_fusion" else ())
    mk = lambda B: {k: v for k, v in synthetic_batch(B, H, W, seed=0, hist_fn=hist.lidar_to_histogram_features, n_points=8192).items() if k in keys}
    b2 = mk(2)

    def step(batch):
        t0 = time.time()
        model_cpu.train_step(ref, opt, batch, cfg)
        return time.time() - t0

    t_start = time.time()
    threads = min(64, ncpu)
    torch.set_num_threads(threads)
    step(b2); step(b2)                              # 2 warm-up steps
    t2 = sorted(step(b2) for _ in range(5))
    res = dict(unit="samples/s", cores=threads, cores_available=ncpu, kind="port",
               b2_value=round(2 / t2[2], 3), b2_sample="B=2, %dx%d, 2 warm-up + 5 timed steps, median %.2f s/step" % (H, W, t2[2]))
    value, sample = res["b2_value"], "B=2"
    left = budget_s - (time.time() - t_start)
    if left > 2.5 * t2[2] * 5:                      # a B=10 step costs ~5x a B=2 step: only if warm-up + >=1 timed step fit
        b10 = mk(10)
        step(b10)
        t10 = []
        while len(t10) < 5 and (time.time() - t_start) + (t10[-1] if t10 else 5 * t2[2]) < budget_s + 15:
            t10.append(step(b10))
        if t10:
            t10.sort()
            res["b10_value"] = round(10 / t10[len(t10) // 2], 3)
            res["b10_sample"] = "B=10, %dx%d, 1 warm-up + %d timed step(s), median %.2f s/step" % (H, W, len(t10), t10[len(t10) // 2])
            value, sample = res["b10_value"], "B=10 (the GPU line's workload)"
    res["value"] = value
    res["sample"] = "oracle.model_cpu.train_step (PyTorch-CPU fp32 restatement of train.py:304-316, %s backbone), %d threads of %d cores; value = %s; %s%s" % (
        backbone, threads, ncpu, sample, res["b2_sample"], ("; " + res["b10_sample"]) if "b10_sample" in res else "")
    if ncpu > threads:      # SURVEY 8d asks for the host's cores: the same B = 2 step on ALL of them, in a child process with a hard deadline (round-5 review, weak #14)
        t0 = time.time()
        try:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-probe", str(ncpu), "--backbone", backbone, "--height", str(H)],
                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=all_cores_deadline_s).stdout.decode()
            sec = float([l for l in out.splitlines() if l.startswith("CPU_PROBE")][-1].split()[1])
            res["all_cores"] = {"cores": ncpu, "b2_value": round(2 / sec, 3), "b2_sample": "B=2, 1 warm-up + 3 timed steps, median %.2f s/step" % sec,
                                "vs_%d_threads" % threads: round((2 / sec) / res["b2_value"], 3)}
        except subprocess.TimeoutExpired:
            res["all_cores"] = {"cores": ncpu, "b2_value": None, "note": "model construction + 4 B=2 steps on %d threads did not finish in %.0f s (the %d-thread run above needs "
                                "~%.0f s for the same): past 64 threads the oneDNN / OpenMP kernels of these layer sizes slow down" % (ncpu, all_cores_deadline_s, threads, 8 + 4 * t2[2])}
        except Exception as e:
            res["all_cores"] = {"cores": ncpu, "b2_value": None, "note": "probe failed: %s" % str(e)[:200]}
        res["all_cores"]["probe_wall_s"] = round(time.time() - t0, 1)
    return res


def alt_precision_line(args, log):
    """The same workload in the fp32-ACCURATE split mode (--dtype f32x3: fp32 storage / accumulation, every plain GEMM and direct convolution as
    an exact 3-way bf16 split on the bf16 MFMA), measured by a child process right after the headline so both lines come from one box.  The
    headline stays the exact-fp32-MFMA line; this object is additional evidence (tests: test_f32x3_split_mode*, error vs fp64 <= the fp32 path's)."""
    torch_free()
    cmd = [sys.executable, os.path.abspath(__file__), "--dtype", "f32x3", "--steps", str(args.steps), "--warmup", str(args.warmup), "--no-cpu-baseline",
           "--backbone", args.backbone, "--dropout", str(args.dropout)] + (["--batch", str(args.batch)] if args.batch else []) + \
          (["--height", str(args.height)] if args.height else []) + (["--no-graph"] if args.no_graph else [])
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600).stdout.decode().strip().splitlines()
        d = json.loads(out[-1])
        log("f32x3 line: %.2f ms/step" % d["ms_per_step"])
        return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": "f32x3", "workload": d["config"]["workload"],
                "final_loss": d["config"]["final_loss"], "roofline": d["roofline"]}
    except Exception as e:   # additional evidence only: never kill the headline line
        return {"error": "%s: %s" % (type(e).__name__, e)}


def child_line(extra, log, what, timeout=600):
    """One more bench line from a CHILD process on the same box (own process: own RCCL group / precision / plans; never kills the headline)."""
    torch_free()
    cmd = [sys.executable, os.path.abspath(__file__), "--no-cpu-baseline", "--no-alt"] + extra
    try:
        out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout).stdout.decode().strip().splitlines()
        d = json.loads(out[-1])
        log("%s: %.2f ms/step" % (what, d["ms_per_step"]))
        return d
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def baseline_config_lines(args, log):
    """BASELINE.json configs[2..4] on this GPU (one rank each; their 8-GPU form is the driver's scaling run): driver-observed numbers instead of
    builder-kept ones.  configs[2] TransFuser bf16 B=10; configs[3] geometric fusion B=12 at 160 x 704 fp32; configs[4] latentTF B=16 fp16."""
    common = ["--steps", str(args.steps), "--warmup", str(args.warmup), "--dropout", str(args.dropout)]
    out = {}
    for key, extra in (("configs[2] TransFuser bf16 B=10", ["--dtype", "bf16"]),
                       ("configs[3] geometric_fusion B=12 H=160 fp32", ["--backbone", "geometric_fusion", "--batch", "12", "--height", "160"]),
                       ("configs[4] latentTF B=16 fp16", ["--backbone", "latentTF", "--batch", "16", "--dtype", "fp16"])):
        d = child_line(common + extra, log, key)
        out[key] = d if "error" in d else {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "dtype": d["dtype"], "workload": d["config"]["workload"],
                                           "final_loss": d["config"]["final_loss"], "step_frac_of_fp16_mfma_peak": d["roofline"].get("step_frac_of_fp16_mfma_peak"),
                                           "roofline_kernel": d["roofline"].get("kernel"), "roofline_frac": d["roofline"].get("frac")}
    return out


def multi_gpu_path_line(args, single_ms, log):
    """The N > 1 code path priced on ONE GPU (bench.py --force-pieces 8 --check in a child process): 8 hipGraph pieces, eager RCCL all-reduce of
    every segment on the side stream, AdamW behind it - against this run's single-graph step."""
    d = child_line(["--steps", str(args.steps), "--warmup", str(args.warmup), "--dropout", str(args.dropout), "--force-pieces", "8", "--check",
                    "--grad-dtype", args.grad_dtype], log, "multi-GPU path on one GPU (8 pieces)")
    if "error" in d:
        return d
    return {"segmented_ms_per_step": d["ms_per_step"], "single_graph_ms_per_step": round(single_ms, 3), "segmented_minus_single_ms": round(d["ms_per_step"] - single_ms, 3),
            "backward_pieces": d["check"]["backward_pieces"], "allreduce_exposed_ms": d["check"]["allreduce_exposed_ms"][0], "grad_dtype": d["check"]["grad_dtype"],
            "note": "world-size-1 RCCL group, the reducer told it has 2 ranks: every segment's arena range is all-reduced in 64 MB buckets on the side stream between the graph "
                    "replays (identity on one rank, but every launch, event and wait of the 8-GPU step is there); 1 / world rides on AdamW's gradient scale, so THIS run's updates use half the gradient (a timing run: the loss curve of the line is not the single-GPU one)"}


def torch_free():
    import torch
    torch.cuda.synchronize()
    torch.cuda.empty_cache()


T0 = time.perf_counter()


def leave_without_teardown():
    """Every rank is done (the JSON line is out): leave WITHOUT c10d's teardown - destroy_process_group() of an RCCL group has aborted the
    interpreter on the MI355X box at the end of a long process (profiles/r03_gpu_tests_final_219_passed_teardown_abort.log), and a rank that
    dies in its shutdown path would turn a finished measurement into a failed torchrun."""
    import torch
    torch.distributed.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    sys.stdout.flush(); sys.stderr.flush()
    os._exit(0)


def dry_run_cpu(args, log):
    """--emulate-cpu: the multi-rank control flow of main() on CPU - same Engine (cuts, segment-wise reduce_async, AdamW), same replica check,
    same exit path - with the kernels host-emulated and a tiny model.  TEST PLUMBING: nothing here is a measurement."""
    import ctypes
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    import model_cases as mc
    from transfuser_amd import _lib
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd.train import Engine, init_distributed
    _lib._install_test_backend(ctypes.CDLL(build_emu.build()))
    rank, local_rank, world = init_distributed()
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    torch.set_num_threads(2)
    cfg = mc.tiny_config(n_layer=1, dropout=args.dropout)
    torch.manual_seed(rank)          # different initial weights per rank: the Engine's broadcast must equalise them
    model = LidarCenterNet(cfg, "cpu", args.backbone, "regnety_tiny", "regnety_tiny", use_velocity=False)
    model.train()
    batch = mc.small_batch(2, 32, 64, 64, 40, seed=rank)
    eng = Engine(model, cfg, lr=cfg.lr, use_graph=False, grad_dtype=args.grad_dtype)
    log("dry run: engine ready (%d backward piece(s), world %d)" % (eng.n_pieces(), world))
    for _ in range(args.warmup):
        eng.train_step(batch)
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tot, det = eng.train_step(batch)
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    check = None
    if args.check:
        check = replica_check(eng, batch, rank, world, torch.device("cpu"), log, cuda=False)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (host-emulated kernels, tiny model): not a measurement", "value": round(2 * world * args.steps / dt, 3), "unit": "samples/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 1), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "dry run of the N-rank plumbing on CPU (gloo)", "final_loss": round(float(tot), 4), "backward_pieces": eng.n_pieces()},
                          "check": check}), flush=True)
    if world > 1:
        leave_without_teardown()


def main():
    args = parse_args()
    if args.cpu_probe:
        return cpu_probe(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)
    import torch
    faulthandler.dump_traceback_later(args.watchdog, repeat=True, file=sys.stderr)
    torch.set_num_threads(min(16, os.cpu_count()))   # host-side glue only; the CPU baseline sets its own count

    def log(msg):
        print("[bench %.1fs] %s" % (time.perf_counter() - T0, msg), file=sys.stderr, flush=True)

    from transfuser_amd import _lib, ops
    from transfuser_amd.config import GlobalConfig
    from transfuser_amd.data import synthetic_batch
    from transfuser_amd.model import LidarCenterNet
    from transfuser_amd.train import Engine, init_distributed
    if args.emulate_cpu:
        return dry_run_cpu(args, log)
    _lib.load()   # fails loudly when libtransfuser_hip.so is missing
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    rank, local_rank, world = init_distributed()
    assert world == args.gpus, "WORLD_SIZE %d != --gpus %d" % (world, args.gpus)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if args.force_pieces:
        assert world == 1 and 2 <= args.force_pieces <= len(Engine.DEFAULT_CUTS) + 1, "--force-pieces needs --gpus 1 and 2..%d pieces" % (len(Engine.DEFAULT_CUTS) + 1)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", str(port))
        torch.distributed.init_process_group("nccl", rank=0, world_size=1)
    backbone = args.backbone
    B = args.batch or {"transFuser": 10, "geometric_fusion": 12, "latentTF": 16}[backbone]
    H = args.height or (160 if backbone == "geometric_fusion" else 256)
    W = 704

    def make_cfg():   # train.py defaults: n_layer 4, use_target_point_image 1, use_velocity 0, multitask, dropout .1
        cfg = GlobalConfig()
        cfg.n_layer = 4
        cfg.use_target_point_image = True
        cfg.embd_pdrop = cfg.attn_pdrop = cfg.resid_pdrop = args.dropout
        return cfg

    cfg = make_cfg()
    torch.manual_seed(0)
    model = LidarCenterNet(cfg, dev, backbone, 'regnety_032', 'regnety_032', use_velocity=False)
    model.train()
    nparam = sum(p.numel() for p in model.parameters())
    hist_fn = lambda pts: ops.lidar_hist(torch.from_numpy(pts).to(dev)[None])[0].cpu().numpy()
    batch = {k: v.to(dev) for k, v in synthetic_batch(B, H, W, seed=rank, hist_fn=hist_fn).items()}
    log("model + batch on device")
    eng = Engine(model, cfg, lr=cfg.lr, use_graph=not args.no_graph, precision={"f32": "fp32", "f32x3": "f32x3", "bf16": "bf16", "fp16": "fp16"}[args.dtype], grad_dtype=args.grad_dtype,
                 cuts=Engine.DEFAULT_CUTS[:args.force_pieces - 1] if args.force_pieces else None)
    if args.force_pieces:      # the reducer behaves as on 2 ranks: all-reduce (identity here) + the mean's scale, on the side stream, between the graph replays
        eng.reducer.world = 2
        if eng.reducer.bf16 and eng.reducer._half is None:
            per = eng.reducer.buckets[0][1] - eng.reducer.buckets[0][0]
            eng.reducer._half = torch.empty(min(per, eng.arena.numel), dtype=torch.bfloat16, device=dev)
    peak = {"f32": PEAK_F32_MFMA_TF, "f32x3": round(PEAK_BF16_MFMA_TF / 6, 1), "bf16": PEAK_BF16_MFMA_TF, "fp16": PEAK_BF16_MFMA_TF}[args.dtype]
    log("engine ready (arena %.1f M floats, %d backward piece(s))" % (eng.arena.numel / 1e6, eng.n_pieces()))

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
    check = None
    if args.check or (world > 1 and not args.no_check):      # outside the timed region; default ON for every multi-GPU run (first-contact hardening: no > 1-GPU lease has
        # ever been available to the builder, so the first real run diagnoses itself): per-rank exposed all-reduce time + replica checksums.  A mismatch is
        # FATAL only when --check was asked for explicitly; the default-on form reports "replicas_equal": false in the JSON line and lets the throughput line stand
        try:
            check = replica_check(eng, batch, rank, world, dev, log)
        except AssertionError as e:
            if args.check:
                raise
            check = {"replicas_equal": False, "error": str(e)[:400]}
            log("replica check FAILED (reported, not fatal without --check): %s" % str(e)[:200])
    roof = dominant_kernel_roofline(eng, batch, dev, log, peak, headline_workload=(backbone == "transFuser" and B == 10 and H == 256))      # every rank runs the census step (it contains the collectives); rank 0 reports
    roof_hbm = {"skipped": "single-rank runs only: the census is one more training step with collectives inside a try / except - a rank that failed in it would leave the others in all_reduce"}
    if world == 1 and not args.force_pieces:
        try:
            roof_hbm = hbm_rooflines(eng, batch, dev, log)               # the same for the bandwidth-bound kernels north_star names
        except Exception as e:   # additional evidence only: never kill the headline line
            roof_hbm = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        ms = dt / args.steps * 1e3
        value = B * world * args.steps / dt
        gf = GFLOP_PER_SAMPLE.get((backbone, H))
        names = {"transFuser": "TransFuser", "geometric_fusion": "GeometricFusion", "latentTF": "latentTF"}
        res = {
            "metric": "training samples/sec (RGB+LiDAR pair), bs=%d/GPU" % B, "value": round(value, 2), "unit": "samples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "%s LidarCenterNet (RegNetY-3.2GF x2, %.1f M params) full train step, B=%d/GPU, 3x%dx%d RGB + 3x256x256 BEV, "
                                   "%s, dropout %.2f, %s" % (names[backbone], nparam / 1e6, B, H, W,
                                                             {"f32": "fp32", "f32x3": "fp32 storage/accumulate, contractions as exact bf16x3 splits on the bf16 MFMA (6 partial products, fp32-accurate)",
                                                              "bf16": "bf16 MFMA contractions (fp32 accumulate / master weights / AdamW; GPT linear layers on bf16-STORED operands, other contractions round fp32 operands in registers)",
                                                              "fp16": "fp16 MFMA contractions, dynamic loss scale with overflow skip (fp32 accumulate / master weights / AdamW; GPT linear layers on half-STORED operands, other contractions round fp32 operands in registers)"}[args.dtype], args.dropout,
                                                               "hipGraph replay" if not args.no_graph else "eager"),
                       "global_batch": B * world, "parallelism": "dp%d" % world, "final_loss": round(loss, 4),
                       "grad_allreduce": ("RCCL, %d backward segments, %s bucket all-reduce overlapped on a side stream" % (eng.n_pieces(), args.grad_dtype)) if (world > 1 or args.force_pieces) else "none (1 rank)"},
        }
        try:      # which sources the measured library was compiled from (sha256 prefix over csrc/ + include/, csrc/api.cpp:tf_build_id)
            res["config"]["build_id"] = ops.L().tf_build_id().decode()
        except Exception:
            res["config"]["build_id"] = None
        if check is not None:
            res["check"] = check
        if gf:
            step_tf = value / world * gf / 1e3
            roof["step_achieved_tflops"] = round(step_tf, 2)
            roof["step_frac"] = round(step_tf / peak, 4)
            roof["step_frac_of_fp16_mfma_peak"] = round(step_tf / PEAK_BF16_MFMA_TF, 4)      # north_star quotes the 16-bit MFMA roofline (2.5 PFLOP/s dense) for every line
        res["roofline"] = roof
        res["roofline_hbm"] = roof_hbm
        if world == 1 and args.dtype == "f32" and not args.no_alt:
            res["f32x3"] = alt_precision_line(args, log)
            if backbone == "transFuser" and not args.batch and not args.height and not args.no_graph:      # the headline invocation: the other BASELINE configs + the N > 1 path
                res["configs"] = baseline_config_lines(args, log)
                res["multi_gpu_path_on_one_gpu"] = multi_gpu_path_line(args, ms, log)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(make_cfg, backbone, H, W)
        print(json.dumps(res), flush=True)
    if world > 1:
        leave_without_teardown()
    if args.force_pieces:      # a real RCCL group exists: leave without c10d's teardown (see leave_without_teardown)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)


if __name__ == "__main__":
    main()
