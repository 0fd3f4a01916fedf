"""Training loop of the hot path: mirror of team_code_transfuser/train.py ``Engine`` (:213-318), the
AdamW optimiser (:142) and the DDP gradient all-reduce (:134), re-designed for MI355X:

* ``ParamArena``   - every parameter and gradient lives in ONE flat fp32 buffer (params keep their
  reference names/shapes as views).  key/query/value of each attention layer are laid out
  back-to-back so the QKV projection is a single N=3C GEMM.  672 MB + 672 MB for TransFuser.
* ``FlatAdamW``    - torch.optim.AdamW semantics in one kernel launch over the arena (K18).
* ``GradReducer``  - data-parallel gradient mean over RCCL (``torch.distributed`` backend "nccl" on
  ROCm; "gloo" in the CPU tests): the arena is cut into large buckets that are all-reduced on a side
  HIP stream while the next bucket is being queued - no per-parameter hooks, no Python in the loop.
* ``Engine``       - ``train_step`` = zero grads -> forward -> weighted loss sum -> backward ->
  (all-reduce) -> AdamW, optionally captured into ONE hipGraph and replayed (the step has ~2.5k
  kernel launches; replay removes the host launch cost and Python entirely).
"""
import os

import torch
import torch.distributed as dist

from . import ops

_ALIGN = 64  # floats (256 B): keeps every parameter 16-byte aligned for float4 / buffer loads


_KQV = __import__("re").compile(r"(^|\.)attn\.(key|query|value)\.")


def _group_key(name):
    """Parameters that must be adjacent: attention key/query/value weights (and biases) per layer."""
    m = _KQV.search(name)
    if m:
        return name[:m.start(2)] + "KQV" + name[m.end(2):], ("key", "query", "value").index(m.group(2))
    return name, 0


class ParamArena:
    def __init__(self, model):
        named, seen = [], set()
        for n, p in model.named_parameters(remove_duplicate=True):
            if id(p) not in seen:
                seen.add(id(p))
                named.append((n, p))
        # parameters the reference's autograd graph never reaches keep grad None there and torch.optim.AdamW skips them
        # (no weight decay either): they go to the END of the arena, outside the range the optimizer updates.
        unused = {id(p) for m in model.modules() if hasattr(m, "unused_parameters") for p in m.unused_parameters()}
        named = [t for t in named if id(t[1]) not in unused] + [t for t in named if id(t[1]) in unused]
        groups, order = {}, []
        for n, p in named:
            k, rank = _group_key(n)
            if k not in groups:
                groups[k] = []
                order.append(k)
            groups[k].append((rank, n, p))
        layout, off = [], 0
        for k in order:
            off = (off + _ALIGN - 1) // _ALIGN * _ALIGN
            for _, n, p in sorted(groups[k], key=lambda t: t[0]):
                layout.append((n, p, off))
                off += p.numel()   # no padding inside a group: members are exactly contiguous
        self.numel = (off + _ALIGN - 1) // _ALIGN * _ALIGN
        first_unused = [o for n, p, o in layout if id(p) in unused]
        self.active_numel = first_unused[0] // _ALIGN * _ALIGN if first_unused else self.numel
        dev = named[0][1].device
        self.params = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.layout = layout
        self.n_params = sum(p.numel() for _, p in named)
        with torch.no_grad():
            for n, p, o in layout:
                # keep each parameter's logical shape AND memory format (channels_last conv weights)
                view = torch.as_strided(self.params, p.shape, p.stride(), o)
                view.copy_(p.data)
                p.data = view
                p.grad = torch.as_strided(self.grads, p.shape, p.stride(), o)

    def zero_grad(self):
        self.grads.zero_()


class FlatAdamW:
    """torch.optim.AdamW(lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01) over a ParamArena.

    ``shard=(rank, world)`` = the reference's ``--zero_redundancy_optimizer 1`` (train.py:143-146, ZeRO stage 1): every rank keeps the
    AdamW moments of - and updates - only its contiguous 1/world slice of the arena; ``GradReducer.all_gather_params`` then circulates
    the updated slices (what ZeroRedundancyOptimizer's parameter broadcast does).  Same parameter trajectory, 1/world of the state."""

    def __init__(self, arena, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, shard=None):
        self.arena = arena
        self.betas, self.eps, self.weight_decay = betas, eps, weight_decay
        n = arena.active_numel
        self.lo, self.hi = 0, n
        if shard is not None and shard[1] > 1:
            rank, world = shard
            per = (n + world - 1) // world
            per = (per + _ALIGN - 1) // _ALIGN * _ALIGN       # slices stay 256-byte aligned
            self.lo, self.hi = min(n, rank * per), min(n, (rank + 1) * per)
            self.shard_size = per
        m = self.hi - self.lo
        self.exp_avg = torch.zeros(m, dtype=torch.float32, device=arena.params.device)
        self.exp_avg_sq = torch.zeros_like(self.exp_avg)
        self.state = torch.tensor([0.0, lr], dtype=torch.float32, device=arena.params.device)  # {step, lr} on the device

    def set_lr(self, lr):
        self.state[1] = lr   # train.py:194-199 (x0.1 at epochs 30 / 40)

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    def step(self):
        a = self.arena
        ops.adamw_(a.params[self.lo:self.hi], a.grads[self.lo:self.hi], self.exp_avg, self.exp_avg_sq, self.state, self.betas[0], self.betas[1],
                   self.eps, self.weight_decay)

    def state_dict(self):
        return dict(exp_avg=self.exp_avg, exp_avg_sq=self.exp_avg_sq, state=self.state)

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"]); self.state.copy_(sd["state"])


class GradReducer:
    """Gradient mean over the data-parallel group (DDP semantics of train.py:134: sum then / world).

    The flat gradient arena is all-reduced in ``bucket_mb`` buckets.  On RCCL the collectives run on
    a side stream, ordered after the backward by an event, and the optimizer waits for the last
    bucket; on MI355X xGMI (7 links/GPU, point-to-point) large buckets keep every link busy."""

    def __init__(self, arena, group=None, bucket_mb=64.0):
        self.arena = arena
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        n = arena.numel
        per = max(_ALIGN, int(bucket_mb * (1 << 20) / 4) // _ALIGN * _ALIGN)
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.stream = torch.cuda.Stream() if arena.grads.is_cuda else None

    def broadcast_params(self, src=0):
        """DDP's initial parameter broadcast (rank 0 -> all)."""
        if self.world > 1:
            dist.broadcast(self.arena.params, src, group=self.group)

    def all_gather_params(self, optimizer):
        """ZeRO-1: after the sharded AdamW step every rank broadcasts its updated slice (rank r owns [r*per, (r+1)*per))."""
        if self.world == 1 or not hasattr(optimizer, "shard_size"):
            return
        p, per, n = self.arena.params, optimizer.shard_size, self.arena.active_numel
        for r in range(self.world):
            lo, hi = min(n, r * per), min(n, (r + 1) * per)
            if hi > lo:
                dist.broadcast(p[lo:hi], r, group=self.group)

    def reduce(self):
        if self.world == 1:
            return
        g = self.arena.grads
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                for s, e in self.buckets:
                    dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=self.group)
                ops.scale_dev_(g, None, None, 1.0 / self.world)
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            for s, e in self.buckets:
                dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=self.group)
            ops.scale_dev_(g, None, None, 1.0 / self.world)


def init_distributed():
    """torchrun env (train.py:94-106): returns (rank, local_rank, world)."""
    if "RANK" not in os.environ or int(os.environ.get("WORLD_SIZE", "1")) == 1:
        return 0, 0, 1
    rank, local_rank, world = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
    backend = "nccl" if torch.cuda.is_available() else "gloo"
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, init_method="env://", world_size=world, rank=rank)
    return rank, local_rank, world


class Engine:
    """One training iteration of train.py:304-316 on a resident batch."""

    BATCH_KEYS = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "bev", "label", "depth", "semantic")
    GEO_KEYS = ("bev_points", "cam_points")   # train.py:280-288 (geometric_fusion only)

    def __init__(self, model, config, lr=1e-4, use_graph=False, group=None, bucket_mb=64.0, wp_only=False, autotune=True, plan_file=None,
                 zero_redundancy_optimizer=False, sync_batch_norm=False):
        self.model = model
        self.config = config
        self._autotune_pending = bool(autotune) and next(model.parameters()).is_cuda
        if plan_file is None:   # tilings tuned offline on an MI355X for the bench / reference shapes (tools/tune.py); unknown shapes are tuned on first use
            plan_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans", "mi355x.txt")
        if next(model.parameters()).is_cuda and os.path.exists(plan_file):
            ops.plans_load(plan_file)
        if sync_batch_norm and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:   # train.py:132-133
            from .functions import convert_sync_batchnorm
            convert_sync_batchnorm(model, group)
            use_graph = False       # collectives between the BatchNorm kernels: eager only
        self.arena = ParamArena(model)
        self.reducer = GradReducer(self.arena, group, bucket_mb)
        self.zero = bool(zero_redundancy_optimizer) and self.reducer.world > 1      # train.py:143-146
        rank = dist.get_rank(group) if self.zero else 0
        self.optimizer = FlatAdamW(self.arena, lr=lr, shard=(rank, self.reducer.world) if self.zero else None)
        self.reducer.broadcast_params()
        w = [1.0] + [0.0] * 10 if wp_only else list(config.detailed_losses_weights)
        self.detailed_weights = dict(zip(config.detailed_losses, w))
        # a head whose loss has weight 0 receives an exactly-zero gradient: its backward kernels are skipped (the gradients stay 0,
        # AdamW still applies weight decay to it, exactly as in the reference)
        loss_to_head = dict(zip(("loss_center_heatmap", "loss_wh", "loss_offset", "loss_yaw_class", "loss_yaw_res", "loss_velocity", "loss_brake"),
                                ("heatmap_head", "wh_head", "offset_head", "yaw_class_head", "yaw_res_head", "velocity_head", "brake_head")))
        model._dead_heads = frozenset(h for k, h in loss_to_head.items() if self.detailed_weights.get(k, 1.0) == 0.0)
        self.use_graph = use_graph
        self._graph = None
        self._static = None
        self._out = None

    def load_data_compute_loss(self, data):
        """train.py:246-293 (transFuser branch); ``data`` tensors must already be on the device."""
        extra = {k: data[k] for k in self.GEO_KEYS} if getattr(self.model, "backbone", "") == "geometric_fusion" else {}
        if getattr(self.model, "use_point_pillars", False):   # train.py:258-260: data["lidar"] is then the raw cloud (B, N, 4)
            extra["num_points"] = data["num_points"]
        return self.model(data["rgb"], data["lidar"], ego_waypoint=data["ego_waypoint"], target_point=data["target_point"],
                          target_point_image=data["target_point_image"], ego_vel=data["ego_vel"].reshape(-1, 1), bev=data["bev"],
                          label=data["label"], depth=data["depth"], semantic=data["semantic"], **extra)

    def _fwd_bwd(self, data):
        if self._autotune_pending and not torch.cuda.is_current_stream_capturing():
            # first eager iteration: let the engine time its candidate tilings for every distinct problem of the step
            self._autotune_pending = False
            ops.autotune(True)
            try:
                return self._fwd_bwd(data)
            finally:
                ops.autotune(False)
        self.optimizer.zero_grad()
        losses = self.load_data_compute_loss(data)
        loss = None
        for key, value in losses.items():   # train.py:307-311
            term = self.detailed_weights[key] * value
            loss = term if loss is None else loss + term
        loss.backward()
        return loss.detach(), {k: v.detach() for k, v in losses.items()}

    def _bump_seed(self):
        seed = getattr(self.model._model, "dropout_seed", None)
        if seed is not None:
            seed.add_(1)

    def train_step(self, data):
        """Returns (total loss, dict of the 11 detailed losses) as device tensors (no host sync)."""
        if not self.use_graph or getattr(self.model, "use_point_pillars", False):   # pillar counts are read on the host: eager only
            out = self._fwd_bwd(data)
            self.reducer.reduce()
            self.optimizer.step()
            if self.zero:
                self.reducer.all_gather_params(self.optimizer)
            self._bump_seed()
            return out
        if self._graph is None:
            self._capture(data)
        else:
            for k in self._static:
                if self._static[k].data_ptr() != data[k].data_ptr():
                    self._static[k].copy_(data[k], non_blocking=True)
        self._graph.replay()
        if self.reducer.world > 1:
            self.reducer.reduce()
            self._opt_graph.replay()
            if self.zero:
                self.reducer.all_gather_params(self.optimizer)
        return self._out

    def _capture(self, data):
        """Capture forward+backward(+AdamW when single-GPU) into a hipGraph.  With >1 rank the gradient
        all-reduce stays outside the graph (eager RCCL calls) followed by a second, tiny AdamW graph."""
        self._static = {k: data[k].clone() for k in self.BATCH_KEYS + self.GEO_KEYS + ("num_points",) if k in data}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):   # warm-up on a side stream (allocator + lazy inits) before capture
            for _ in range(2):
                self._fwd_bwd(self._static)
                self.reducer.reduce()
                self.optimizer.step()
                if self.zero:
                    self.reducer.all_gather_params(self.optimizer)
                self._bump_seed()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        single = self.reducer.world == 1
        with torch.cuda.graph(self._graph):
            self._out = self._fwd_bwd(self._static)
            if single:
                self.optimizer.step()
                self._bump_seed()
        if not single:
            self._opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._opt_graph):
                self.optimizer.step()
                self._bump_seed()

    def save(self, path_prefix, epoch):
        """train.py:381-384 (rank 0)."""
        torch.save(self.model.state_dict(), "%s/model_%d.pth" % (path_prefix, epoch))
        torch.save(self.optimizer.state_dict(), "%s/optimizer_%d.pth" % (path_prefix, epoch))
