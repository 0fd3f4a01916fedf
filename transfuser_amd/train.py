"""Training loop of the hot path: mirror of team_code_transfuser/train.py ``Engine`` (:213-318), the
AdamW optimiser (:142) and the DDP gradient all-reduce (:134), re-designed for MI355X:

* ``ParamArena``   - every parameter and gradient lives in ONE flat fp32 buffer (params keep their
  reference names/shapes as views).  key/query/value of each attention layer are laid out
  back-to-back so the QKV projection is a single N=3C GEMM.  672 MB + 672 MB for TransFuser.
* ``FlatAdamW``    - torch.optim.AdamW semantics in one kernel launch over the arena (K18).
* ``GradReducer``  - data-parallel gradient mean over RCCL (``torch.distributed`` backend "nccl" on
  ROCm; "gloo" in the CPU tests): the arena is cut into large buckets that are all-reduced on a side
  HIP stream - no per-parameter hooks, no Python in the loop.  ``reduce_async(lo, hi)`` starts the
  all-reduce of one arena range as soon as the backward SEGMENT that produces it has been enqueued.
* ``Engine``       - ``train_step`` = zero grads -> forward -> weighted loss sum -> backward ->
  (all-reduce) -> AdamW, optionally captured into hipGraphs and replayed (the step has ~2.5k
  kernel launches; replay removes the host launch cost and Python entirely).  With more than one
  rank the backward is cut at the fusion-stage boundaries into segments (one graph each) and the
  arena is laid out in backward-ready order, so the all-reduce of the ~130 M stage-4 / head gradients
  runs on the side stream while stages 3..1 are still being differentiated (DDP's overlap,
  train.py:134, without hooks).
"""
import os

import torch
import torch.distributed as dist

from . import ops
from . import functions as _F

_ALIGN = 64  # floats (256 B): keeps every parameter 16-byte aligned for float4 / buffer loads


_KQV = __import__("re").compile(r"(^|\.)attn\.(key|query|value)\.")


_HEAD3 = __import__("re").compile(r"(?:^|\.)(?:head\.(\w+_head)|(pred_bev))\.0\.(weight|bias)$")


def _group_key(name):
    """Parameters that must be adjacent: attention key/query/value weights (and biases) per layer; the first (3x3) convolutions of the seven
    CenterNet heads and pred_bev in model.MERGED_ORDER (one 64 -> 512 convolution, model.merged_head_convs)."""
    m = _KQV.search(name)
    if m:
        return name[:m.start(2)] + "KQV" + name[m.end(2):], ("key", "query", "value").index(m.group(2))
    m = _HEAD3.search(name)
    if m:
        from .model import MERGED_ORDER
        which = m.group(1) or m.group(2)
        if which in MERGED_ORDER:
            return name[:m.start()] + "HEADS3x3." + m.group(3), MERGED_ORDER.index(which)
    return name, 0


_STAGE_RE = __import__("re").compile(r"(?:^|\.)(?:image_encoder\.features|lidar_encoder\._model)\.(?:s|layer)([1-4])\.|(?:^|\.)transformer([1-4])\.(?:(blocks)\.(\d+)\.|(ln_f)\.)?")
_CNX_STAGE_RE = __import__("re").compile(r"(?:^|\.)(?:image_encoder\.features|lidar_encoder\._model)\.stages\.([0-3])\.")      # ConvNeXt: stages.i is aliased as layer{i+1}
_STEM_RE = __import__("re").compile(r"(?:^|\.)(?:image_encoder\.features|lidar_encoder\._model)\.(?:stem|conv1|bn1)\.|(?:^|\.)point_pillar_net\.")
_LAST = 1 << 20


def param_key(name):
    """Where in the forward pass a parameter is used, as a tuple that sorts in forward order - the same keys the backbone's cut points
    use (transfuser._FusionBackbone._cuts): (i, 0, 0) = RegNet stage i of either trunk; (i, 1, 0) = GPT i's embedding (pos_emb, vel_emb);
    (i, 1, j + 1) = Block j of GPT i; (i, 1, _LAST) = its ln_f; (0, 0, 0) = the two stems AND everything upstream of them (the PointPillars
    point net, model.py:736-738: its gradients are produced by the very LAST backward piece, through the LiDAR stem); (5, 0, 0) = everything
    after the backbone's last stage (channel reducers, FPN, decoders, heads, join / GRU).  The backward produces gradients in DEcreasing key
    order, and a segment's arena range is all-reduced as soon as its piece is enqueued - a parameter filed later than the point that
    produces its gradient would be reduced while still zero and then diverge between the replicas."""
    m = _CNX_STAGE_RE.search(name)
    if m:
        return (int(m.group(1)) + 1, 0, 0)
    m = _STAGE_RE.search(name)
    if m:
        if m.group(1):
            return (int(m.group(1)), 0, 0)
        i = int(m.group(2))
        if m.group(3):
            return (i, 1, int(m.group(4)) + 1)
        return (i, 1, _LAST if m.group(5) else 0)
    return (0, 0, 0) if _STEM_RE.search(name) else (5, 0, 0)


def param_stage(name):
    """Fusion stage of a parameter (first component of param_key): 1..4 = RegNet stage k of either trunk / GPT k, 0 = stems (+ point net), 5 = heads."""
    return param_key(name)[0]


def cut_key(c):
    """A backward cut as a key tuple; an int c means "after fusion stage c" = (c, 2, 0)."""
    return (int(c), 2, 0) if isinstance(c, int) else tuple(int(v) for v in c)


class ParamArena:
    def __init__(self, model, cuts=()):
        """``cuts`` = points at which the backward is cut (cut_key: ints = after fusion stage c, tuples = the backbone's finer cut points,
        e.g. (4, 1, 2) = inside GPT-4 in front of Block 2): parameters are then grouped by backward segment, first-finished segment first, and
        ``segment_ranges`` lists each segment's [lo, hi) float range of the arena."""
        named, seen = [], set()
        for n, p in model.named_parameters(remove_duplicate=True):
            if id(p) not in seen:
                seen.add(id(p))
                named.append((n, p))
        # parameters the reference's autograd graph never reaches keep grad None there and torch.optim.AdamW skips them
        # (no weight decay either): they go to the END of the arena, outside the range the optimizer updates.
        unused = {id(p) for m in model.modules() if hasattr(m, "unused_parameters") for p in m.unused_parameters()}
        named = [t for t in named if id(t[1]) not in unused] + [t for t in named if id(t[1]) in unused]
        cuts = sorted(set(cut_key(c) for c in cuts), reverse=True)
        seg_of = lambda name: sum(1 for c in cuts if param_key(name) <= c)       # 0 = produced first in the backward
        if cuts:   # stable sort by segment; the never-reached parameters stay at the very end
            live = [t for t in named if id(t[1]) not in unused]
            named = sorted(live, key=lambda t: seg_of(t[0])) + [t for t in named if id(t[1]) in unused]
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
        # [lo, hi) of every backward segment (aligned; together they cover [0, active_numel)); one range when there are no cuts
        nseg = len(cuts) + 1
        first = [None] * nseg
        for n, p, o in layout:
            if id(p) not in unused and first[seg_of(n)] is None:
                first[seg_of(n)] = o // _ALIGN * _ALIGN
        self.segment_ranges, hi = [], self.active_numel
        for k in reversed(range(nseg)):
            lo = first[k] if first[k] is not None else hi
            self.segment_ranges.insert(0, (lo, hi))
            hi = lo
        assert self.segment_ranges[0][0] == 0 or not cuts
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
        ls = getattr(self, "ls_state", None)
        if ls is not None:      # fp16 mode with the dynamic loss scale: overflow check over the WHOLE (reduced) gradient arena, skip-or-update, scale update
            ops.adamw_dynscale_(a.params[self.lo:self.hi], a.grads[self.lo:self.hi], self.exp_avg, self.exp_avg_sq, self.state, ls, a.grads[:a.active_numel],
                                self.betas[0], self.betas[1], self.eps, self.weight_decay)
            return
        gs = getattr(self, "grad_scale", 1.0)
        red = getattr(self, "reducer", None)
        if red is not None:      # defer_scale: the arena holds the all-reduced SUM, the mean's 1 / world rides on the gradient scale (GradReducer.grad_scale)
            gs = gs * red.grad_scale()
        ops.adamw_(a.params[self.lo:self.hi], a.grads[self.lo:self.hi], self.exp_avg, self.exp_avg_sq, self.state, self.betas[0], self.betas[1],
                   self.eps, self.weight_decay, grad_scale=gs)

    def _layout(self):
        """[(parameter name, arena offset, numel)] of the parameters the optimizer updates: what makes a saved state independent of the
        arena ORDER (which depends on the backward cuts, i.e. on the world size the run used)."""
        n = self.arena.active_numel
        return [(name, off, p.numel()) for name, p, off in self.arena.layout if off < n]

    def state_dict(self, group=None):
        """FULL optimizer state (moments over the whole active arena, {step, lr}) plus the (name, offset, numel) layout they were written
        in.  With a sharded optimizer every rank must call this (collective): the slices are gathered first - the reference's
        ``optimizer.consolidate_state_dict(0)`` before saving (train.py:206-207).  ``load_state_dict`` re-maps the moments PER PARAMETER
        NAME, so a checkpoint depends neither on the world size nor on the backward cuts it was written with."""
        ea, es = self.exp_avg, self.exp_avg_sq
        if hasattr(self, "shard_size") and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            world, per, n = dist.get_world_size(group), self.shard_size, self.arena.active_numel
            full = []
            for t in (ea, es):
                mine = torch.zeros(per, dtype=t.dtype, device=t.device)
                mine[:t.numel()] = t
                parts = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(parts, mine, group=group)
                full.append(torch.cat(parts)[:n].clone())
            ea, es = full
        sd = dict(exp_avg=ea, exp_avg_sq=es, state=self.state, active_numel=self.arena.active_numel, layout=self._layout())
        if getattr(self, "ls_state", None) is not None:      # fp16 mode: {scale, clean steps, found_inf, growth interval} - a resumed run continues the scale schedule
            sd["ls_state"] = self.ls_state
        return sd

    def load_state_dict(self, sd):
        """Accepts the full state written by ``state_dict`` on any world size / with any backward cuts: the moments are copied parameter
        by parameter (matched by name); a sharded optimizer keeps its own slice.  A state without a layout (written before the layout
        was recorded) is only accepted when this arena has no cuts - its order is then the one those files were written in on one GPU;
        anything else raises instead of silently assigning moments to the wrong parameters."""
        n = self.arena.active_numel
        mine = self._layout()
        theirs = sd.get("layout")
        if theirs is None:
            if len(self.arena.segment_ranges) > 1:
                raise ValueError("optimizer state has no parameter layout (written by an older build) and this arena is laid out for %d backward "
                                 "segments: the flat order cannot be matched - re-save the checkpoint with the current build" % len(self.arena.segment_ranges))
            theirs = mine
        theirs = [tuple(t) for t in theirs]
        same = theirs == [tuple(t) for t in mine]
        if not same:
            a, b = {t[0]: t for t in theirs}, {t[0]: t for t in mine}
            missing = [k for k in b if k not in a or a[k][2] != b[k][2]]
            if missing or len(a) != len(b):
                raise ValueError("optimizer state does not match this model: %d parameters differ (e.g. %s)" % (max(len(missing), abs(len(a) - len(b))), missing[:3]))
        for name, dst in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
            src = sd[name].to(dst.device)
            if src.numel() != sd.get("active_numel", n):
                raise ValueError("optimizer state %s has %d elements, its header says %d" % (name, src.numel(), sd.get("active_numel", n)))
            if same:
                full = src
            else:           # re-map by parameter name into this arena's order (padding between groups stays zero)
                full = torch.zeros(n, dtype=dst.dtype, device=dst.device)
                off_theirs = {t[0]: t[1] for t in theirs}
                for pname, off, cnt in mine:
                    full[off:off + cnt] = src[off_theirs[pname]:off_theirs[pname] + cnt]
            if full.numel() != n:
                raise ValueError("optimizer state %s has %d elements, expected %d" % (name, full.numel(), n))
            dst.copy_(full[self.lo:self.hi] if dst.numel() != n else full)
        self.state.copy_(sd["state"])
        if sd.get("ls_state") is not None and getattr(self, "ls_state", None) is not None:
            self.ls_state.copy_(sd["ls_state"].to(self.ls_state.device))      # in place: the Engine's backward seed is a view of this tensor


class GradReducer:
    """Gradient mean over the data-parallel group (DDP semantics of train.py:134: sum then / world).

    The flat gradient arena is all-reduced in ``bucket_mb`` buckets.  On RCCL the collectives run on
    a side stream, ordered after the backward by an event, and the optimizer waits for the last
    bucket; on MI355X xGMI (7 links/GPU, point-to-point) large buckets keep every link busy.
    ``grad_dtype="bf16"`` (opt-in; the reference reduces fp32): every bucket is rounded to bf16 on its way out and widened (x 1/world) on
    its way back, halving the xGMI payload (336 MB instead of 672 MB per step); the arena itself stays fp32."""

    def __init__(self, arena, group=None, bucket_mb=64.0, grad_dtype="fp32"):
        self.arena = arena
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        n = arena.numel
        per = max(_ALIGN, int(bucket_mb * (1 << 20) / 4) // _ALIGN * _ALIGN)
        self.buckets = [(s, min(n, s + per)) for s in range(0, n, per)]
        self.stream = torch.cuda.Stream() if arena.grads.is_cuda else None
        assert grad_dtype in ("fp32", "bf16"), grad_dtype
        self.bf16 = grad_dtype == "bf16"
        self._half = torch.empty(min(per, n), dtype=torch.bfloat16, device=arena.grads.device) if self.bf16 and self.world > 1 else None
        self.defer_scale = False       # fp32 buckets: leave the SUM in the arena, the optimizer multiplies by 1 / world (train.Engine sets it when its AdamW can)
        self._wait_events = None       # (before, after) timing events around the optimizer's wait for the side stream: the EXPOSED all-reduce time
        self.time_waits = False        # bench.py --check turns the event pair on

    def grad_scale(self):
        """The factor that turns the gradient arena (and every ``p.grad`` view of it) into the MEAN gradient after ``train_step``: with ``defer_scale``
        (fp32 buckets, world > 1, no dynamic loss scale) the arena holds the all-reduced SUM until the optimizer step and FlatAdamW applies 1 / world
        on its way in - anything ELSE that reads gradients (norm logging / clipping, a user optimizer, a checksum) must multiply by this.  1.0 otherwise."""
        return 1.0 / self.world if (self.defer_scale and self.world > 1) else 1.0

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

    def _reduce_range(self, lo, hi):
        """sum over the ranks, then x 1/world, of grads[lo:hi], bucket by bucket on the current stream."""
        g = self.arena.grads
        per = self.buckets[0][1] - self.buckets[0][0]
        for s in range(lo, hi, per):
            e = min(hi, s + per)
            if self.bf16:
                h = self._half[:e - s]
                ops.cast_bf16(g[s:e], out=h)
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                ops.cast_f32(h, g[s:e], 1.0 / self.world)
            else:
                dist.all_reduce(g[s:e], op=dist.ReduceOp.SUM, group=self.group)
                if not self.defer_scale:      # otherwise AdamW applies 1 / world on its way in (its gradient scale): no launch per bucket
                    ops.scale_dev_(g[s:e], None, None, 1.0 / self.world)

    def reduce_async(self, lo, hi):
        """All-reduce (mean) the gradient range [lo, hi) on the side stream, ordered after everything enqueued so far on the current
        stream - call it right after the backward segment that produces the range; the current stream keeps running the next segment."""
        if self.world == 1 or hi <= lo:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._reduce_range(lo, hi)
        else:
            self._reduce_range(lo, hi)

    def finish(self):
        """The optimizer (current stream) waits for every pending bucket."""
        if self.world > 1 and self.stream is not None:
            cur = torch.cuda.current_stream()
            if self.time_waits:
                ev = self._wait_events = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record(cur)
                cur.wait_stream(self.stream)
                ev[1].record(cur)
            else:
                cur.wait_stream(self.stream)

    def exposed_ms(self):
        """Time the current stream stalled in the last ``finish()`` = the part of the all-reduce the backward did NOT hide (needs
        ``time_waits``; synchronises).  0.0 on one rank."""
        if self._wait_events is None:
            return 0.0
        self._wait_events[1].synchronize()
        return float(self._wait_events[0].elapsed_time(self._wait_events[1]))

    def reduce(self):
        if self.world == 1:
            return
        if self.stream is not None:
            self.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self._reduce_range(0, self.arena.numel)
            torch.cuda.current_stream().wait_stream(self.stream)
        else:
            self._reduce_range(0, self.arena.numel)


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
    # backward pieces with more than one rank (8): [heads + neck + GPT-4 ln_f / Block 3] [GPT-4 Block 2] [Block 1] [Block 0 + embedding +
    # stage-4 trunks] [GPT-3] [stage-3 trunks] [stage 2] [stage 1 + stems].  GPT-4 holds 110 M of the 168 M gradients (27.5 M per Block):
    # cutting between its Blocks lets the first 190 MB leave for the all-reduce after ~15 % of the backward instead of after stage 4.
    DEFAULT_CUTS = ((4, 1, 3), (4, 1, 2), (4, 1, 1), 3, (3, 0, 0), 2, 1)

    def __init__(self, model, config, lr=1e-4, use_graph=False, group=None, bucket_mb=64.0, wp_only=False, autotune=True, plan_file=None,
                 zero_redundancy_optimizer=False, sync_batch_norm=False, cuts=None, precision=None, grad_dtype="fp32", loss_scale=None):
        """``cuts``: points (cut_key: int c = after fusion stage c; (i, 0, 0) = between the stage-i trunks and GPT i; (i, 1, j) = inside GPT i
        in front of Block j) at which the backward is cut into separately enqueued (and separately captured) segments whose gradient
        ranges are all-reduced while the next segment runs.  None = DEFAULT_CUTS when there is more than one rank (and the backbone is a
        chain: transFuser / latentTF), no cut on a single GPU; () = never cut.
        ``precision``: compute precision of every MFMA-engine contraction (process-wide, ops.set_precision); None keeps the current one.
        ``loss_scale``: a number = STATIC loss scale S (the backward is seeded with S instead of 1, AdamW multiplies the gradients by 1 / S);
        None = 1 in the fp32 / f32x3 / bf16 modes and a DYNAMIC scale in fp16 mode (initial 2^16, device-side overflow check in front of
        AdamW: the step is skipped and the scale halved on Inf / NaN gradients, doubled after ``loss_scale_growth_interval`` = 2000 clean
        steps - torch.cuda.amp.GradScaler's policy; the reference trains fp32 and has no counterpart); formerly None = static 1024
        in "fp16" precision (half operands flush gradients below 6e-8 to zero) and 1 otherwise.  The reported losses are unscaled."""
        self.model = model
        self.config = config
        if precision is not None:   # "fp32" (exact fp32 MFMA: the reference's arithmetic) | "f32x3" (bf16x3 split on the bf16 MFMA, fp32-accurate) | "bf16" (bf16 MFMA operands, fp32 accumulate / storage / master weights)
            ops.set_precision(precision)
        self._autotune_pending = bool(autotune) and next(model.parameters()).is_cuda
        if plan_file is None:   # tilings tuned offline on an MI355X for the bench / reference shapes (tools/tune.py); unknown shapes are tuned on first use
            plan_file = os.environ.get("TF_PLANS") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "plans", "mi355x.txt")
        if next(model.parameters()).is_cuda and os.path.exists(plan_file):
            ops.plans_load(plan_file)
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        if sync_batch_norm and world > 1:   # train.py:132-133
            from .functions import convert_sync_batchnorm
            convert_sync_batchnorm(model, group)
            use_graph = False       # collectives between the BatchNorm kernels: eager only
        backbone = getattr(model, "_model", None)
        can_cut = getattr(model, "backbone", "") in ("transFuser", "latentTF") and hasattr(backbone, "_cuts")
        if cuts is None:
            cuts = self.DEFAULT_CUTS if world > 1 else ()
        nblk = int(getattr(config, "n_layer", 0))
        keys = sorted(set(cut_key(c) for c in cuts), reverse=True) if can_cut else []
        self.cuts = tuple(k for k in keys if not (k[1] == 1 and k[2] >= nblk))       # a cut in front of a Block the model does not have is dropped
        if can_cut:
            backbone._cuts = frozenset(self.cuts)
        self.arena = ParamArena(model, self.cuts)
        self.reducer = GradReducer(self.arena, group, bucket_mb, grad_dtype)
        self.zero = bool(zero_redundancy_optimizer) and self.reducer.world > 1      # train.py:143-146
        rank = dist.get_rank(group) if self.zero else 0
        self.optimizer = FlatAdamW(self.arena, lr=lr, shard=(rank, self.reducer.world) if self.zero else None)
        self.ls_state = None
        if loss_scale is None and ops.get_precision() == "fp16":
            dev = self.arena.params.device
            self.ls_state = torch.tensor([65536.0, 0.0, 0.0, float(getattr(config, "loss_scale_growth_interval", 2000))], dtype=torch.float32, device=dev)
            self.optimizer.ls_state = self.ls_state
            loss_scale = 65536.0
        self.loss_scale = float(loss_scale if loss_scale is not None else 1.0)
        self.optimizer.grad_scale = 1.0 / self.loss_scale
        # fp32 gradient buckets: no 1 / world launch per bucket behind the all-reduce, AdamW's gradient scale carries it (the dynamic loss scale
        # computes its own scale on the device and keeps the per-bucket launch; bf16 buckets scale for free while they are widened)
        self.optimizer.reducer = self.reducer
        self.reducer.defer_scale = self.ls_state is None and not self.reducer.bf16
        self._seed_grad = self.ls_state[0] if self.ls_state is not None else None      # a VIEW of the device scale: the update kernel re-seeds the next backward
        self.reducer.broadcast_params()
        w = [1.0] + [0.0] * 10 if wp_only else list(config.detailed_losses_weights)
        self.detailed_weights = dict(zip(config.detailed_losses, w))
        # a head whose loss has weight 0 receives an exactly-zero gradient: its backward kernels are skipped (the gradients stay 0,
        # AdamW still applies weight decay to it, exactly as in the reference)
        loss_to_head = dict(zip(("loss_center_heatmap", "loss_wh", "loss_offset", "loss_yaw_class", "loss_yaw_res", "loss_velocity", "loss_brake"),
                                ("heatmap_head", "wh_head", "offset_head", "yaw_class_head", "yaw_res_head", "velocity_head", "brake_head")))
        model._dead_heads = frozenset(h for k, h in loss_to_head.items() if self.detailed_weights.get(k, 1.0) == 0.0)
        self.use_graph = use_graph
        ppn = getattr(model, "point_pillar_net", None) if getattr(model, "use_point_pillars", False) else None
        if ppn is not None:      # under a captured graph the PointPillars front-end runs with static shapes (no host read of the kept-point / pillar counts)
            ppn.static_shapes = bool(use_graph)
        self._graphs = None
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

    # ---- the step as a sequence of PIECES: piece 0 = zero grads + forward + weighted loss + backward down to the first cut, piece i > 0 =
    # backward of the next segment (restarted from the detached boundary tensors the backbone recorded).  After piece i the arena range
    # self.arena.segment_ranges[i] holds this rank's final gradients.  Eager mode runs the pieces back to back; graph mode captures one
    # hipGraph per piece; in both the reducer is told about a range as soon as its piece is enqueued.
    def n_pieces(self):
        return len(self.cuts) + 1

    def _piece0(self, data):
        with _F.inplace_param_grads(), ops.lowp_managed():
            return self._piece0_impl(data)

    def _piece(self, i):
        with _F.inplace_param_grads(), ops.lowp_managed():
            return self._piece_impl(i)

    def _opt_step(self):
        """AdamW over the arena, then the 16-bit weight copies of the storage modes are rewritten from the updated master weights.  With
        ZeRO-1 a rank only holds ITS shard's update at this point: the copies are rewritten after the parameter all-gather instead
        (``_after_opt``), otherwise the cached copies of the other ranks' shards would lag one step and differ from rank to rank."""
        self.optimizer.step()
        if ops.lowp_storage() and not self.zero:
            ops.lowp_refresh_weights()

    def _after_opt(self):
        """Eager tail of a step behind AdamW (outside the captured graphs: the all-gather is an RCCL call)."""
        if self.zero:
            self.reducer.all_gather_params(self.optimizer)
            if ops.lowp_storage():
                ops.lowp_refresh_weights()

    def _piece0_impl(self, data):
        self.optimizer.zero_grad()
        losses = self.load_data_compute_loss(data)
        keys = list(losses)                  # train.py:307-311: loss = sum of weight * detailed loss, in dictionary order - one launch per direction
        if ops._AB_WSUM and 1 <= len(keys) <= 16 and all(v.dim() == 0 and v.dtype == torch.float32 and v.is_contiguous() for v in losses.values()):
            loss = _F.WeightedSumFn.apply(tuple(self.detailed_weights[k] for k in keys), *[losses[k] for k in keys])
        else:
            loss = None
            for key, value in losses.items():
                term = self.detailed_weights[key] * value
                loss = term if loss is None else loss + term
        if self.ls_state is not None:
            loss.backward(self._seed_grad)
        elif self.loss_scale != 1.0:
            if self._seed_grad is None or self._seed_grad.device != loss.device:
                self._seed_grad = torch.full((), self.loss_scale, dtype=torch.float32, device=loss.device)
            loss.backward(self._seed_grad)
        else:
            loss.backward()
        self._pending = list(getattr(getattr(self.model, "_model", None), "_boundaries", ()) or ()) if self.cuts else []
        assert len(self._pending) == len(self.cuts), "backbone recorded %d boundaries for %d cuts" % (len(self._pending), len(self.cuts))
        return loss.detach(), {k: v.detach() for k, v in losses.items()}

    def _piece_impl(self, i):
        outs, leaves = self._pending[len(self._pending) - i]        # boundaries were recorded in forward order; backward walks them in reverse
        pairs = [(o, l.grad) for o, l in zip(outs, leaves) if l.grad is not None]      # a boundary tensor nothing downstream differentiated has no gradient
        if pairs:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        if i == len(self._pending):
            self._pending = []
            self.model._model._boundaries = []

    def _fwd_bwd(self, data, reduce=False):
        if self._autotune_pending and not torch.cuda.is_current_stream_capturing():
            # first eager iteration: let the engine time its candidate tilings for every distinct problem of the step
            self._autotune_pending = False
            ops.autotune(True)
            try:
                return self._fwd_bwd(data, reduce)
            finally:
                ops.autotune(False)
        out = self._piece0(data)
        if reduce:
            self.reducer.reduce_async(*self.arena.segment_ranges[0])
        for i in range(1, self.n_pieces()):
            self._piece(i)
            if reduce:
                self.reducer.reduce_async(*self.arena.segment_ranges[i])
        if reduce:
            self.reducer.finish()
        return out

    def _bump_seed(self):
        seed = getattr(self.model._model, "dropout_seed", None)
        if seed is not None:
            seed.add_(1)

    def _eager_step(self, data):
        out = self._fwd_bwd(data, reduce=True)
        self._opt_step()
        self._after_opt()
        self._bump_seed()
        return out

    def train_step(self, data):
        """Returns (total loss, dict of the 11 detailed losses) as device tensors (no host sync)."""
        if not self.use_graph:
            return self._eager_step(data)
        if self._graphs is None:
            self._capture(data)
        else:
            for k in self._static:
                if self._static[k].data_ptr() != data[k].data_ptr():
                    self._static[k].copy_(data[k], non_blocking=True)
        for i, g in enumerate(self._graphs):
            g.replay()
            self.reducer.reduce_async(*self.arena.segment_ranges[i])      # no-op on a single rank
        if self._opt_graph is not None:
            self.reducer.finish()
            self._opt_graph.replay()
            self._after_opt()
        return self._out

    def _capture(self, data):
        """Capture the step into hipGraphs: ONE graph (forward + backward + AdamW) on a single GPU without cuts; otherwise one graph per
        backward piece (sharing a memory pool; the autograd graph built while capturing piece 0 is consumed by the later captures) plus a
        tiny AdamW graph - the gradient all-reduces stay outside the graphs as eager RCCL calls on the reducer's side stream."""
        self._static = {k: data[k].clone() for k in self.BATCH_KEYS + self.GEO_KEYS + ("num_points",) if k in data}
        # The warm-up iterations (allocator, lazy inits, autotune) must leave NO trace: the first replay is training step 1, exactly as in
        # the eager loop / the reference's one-update-per-batch loop (train.py:304-316).  Everything a step mutates is snapshotted and put
        # back: parameters, AdamW moments + step counter, BatchNorm running statistics / dropout seed (all module buffers).
        opt = self.optimizer
        snap = [(t, t.clone()) for t in [self.arena.params, opt.exp_avg, opt.exp_avg_sq, opt.state] + ([self.ls_state] if self.ls_state is not None else []) + list(self.model.buffers())]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):   # warm-up on a side stream (allocator + lazy inits) before capture
            for _ in range(2):
                self._eager_step(self._static)
            with torch.no_grad():
                for t, saved in snap:
                    t.copy_(saved)
            if ops.lowp_storage():      # the cached 16-bit weight copies were last written from the warm-up's parameters: re-make them from the restored ones
                ops.lowp_refresh_weights()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        del snap
        fused_opt = self.reducer.world == 1 and self.n_pieces() == 1
        self._graphs = [torch.cuda.CUDAGraph() for _ in range(self.n_pieces())]
        # With a process group alive, c10d's watchdog THREAD polls the events of the (already finished) warm-up collectives; in the default
        # "global" capture mode such a query from another thread is an error that kills the process ("operation not permitted when stream is
        # capturing", seen on the MI355X with a world-size-1 RCCL group).  Thread-local mode restricts the check to the capturing thread.
        mode = dict(capture_error_mode="thread_local") if (dist.is_available() and dist.is_initialized()) else {}
        with torch.cuda.graph(self._graphs[0], **mode):
            self._out = self._piece0(self._static)
            if fused_opt:
                self._opt_step()
                self._bump_seed()
        pool = self._graphs[0].pool()
        for i in range(1, self.n_pieces()):
            with torch.cuda.graph(self._graphs[i], pool=pool, **mode):
                self._piece(i)
        if fused_opt:
            self._opt_graph = None
        else:
            self._opt_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._opt_graph, pool=pool, **mode):
                self._opt_step()
                self._bump_seed()

    def save(self, path_prefix, epoch):
        """train.py:204-210,381-384: every rank calls it (the ZeRO state is consolidated collectively), rank 0 writes."""
        osd = self.optimizer.state_dict(self.reducer.group)
        if self.reducer.world == 1 or dist.get_rank(self.reducer.group) == 0:
            torch.save(self.model.state_dict(), "%s/model_%d.pth" % (path_prefix, epoch))
            torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}, "%s/optimizer_%d.pth" % (path_prefix, epoch))


# ================================================================================================ command line (train.py:27-211, 295-384)
def seed_worker(worker_id):
    """train.py:386-391."""
    import random
    import numpy as np
    worker_seed = torch.initial_seed() % 2 ** 32
    np.random.seed(worker_seed)
    random.seed(worker_seed)


class Trainer:
    """The reference's epoch-level ``Engine`` (train.py:213-384) around the step ``Engine`` above: ``train()`` one epoch over the loader
    (H2D, step, running sums), ``validate()`` (eval mode, inference_mode, same weighted sums), ``log_losses`` (averages gathered on
    rank 0 with ``gather_object``; TensorBoard scalars when tensorboard is importable, always ``losses.jsonl``), ``save`` (model_%d.pth +
    optimizer_%d.pth, DDP's ``module.`` key prefix under parallel training for checkpoint interchange with the reference agent)."""

    def __init__(self, step_engine, dataloader_train, dataloader_val, args, config, device, rank=0, world_size=1, parallel=False, cur_epoch=0):
        self.eng, self.dataloader_train, self.dataloader_val = step_engine, dataloader_train, dataloader_val
        self.args, self.config, self.device, self.rank, self.world_size, self.parallel = args, config, device, rank, world_size, parallel
        self.cur_epoch = cur_epoch
        self.detailed_losses = config.detailed_losses
        self.writer = None
        if rank == 0:
            try:
                from torch.utils.tensorboard import SummaryWriter
                self.writer = SummaryWriter(log_dir=args.logdir)
            except Exception:   # tensorboard is optional here; the JSONL log below always exists
                self.writer = None

    def _to_device(self, data):
        """train.py:246-271: H2D + dtype casts of one collated batch.  A RAW batch (``CARLA_Data`` items: uint8 images, the padded cloud,
        poses) first goes through ``GpuBatchPrep``: one H2D copy, then alignment / histogram / crops / decoding in HIP kernels."""
        if "rgb_u8" in data:
            if getattr(self, "_prep", None) is None:
                from .data import GpuBatchPrep
                self._prep = GpuBatchPrep(self.config, self.device, correspondences=getattr(self.args, "backbone", "") == "geometric_fusion",
                                          seed=1000003 * self.rank)
            data = self._prep(data)                 # with --use_point_pillars data["lidar"] is then the aligned cloud (train.py:258-260) + num_points
        f32 = ("rgb", "lidar", "ego_waypoint", "target_point", "target_point_image", "ego_vel", "label", "depth")
        i64 = ("bev", "semantic", "bev_points", "cam_points")
        out = {}
        for k, v in data.items():
            if k in f32:
                out[k] = v.to(self.device, dtype=torch.float32, non_blocking=True)
            elif k in i64:
                out[k] = v.to(self.device, dtype=torch.long, non_blocking=True)
            elif k == "num_points":
                out[k] = v.to(self.device, dtype=torch.int32, non_blocking=True)
        return out

    def train(self):
        self.eng.model.train()
        num_batches, loss_epoch = 0, torch.zeros((), device=self.device)
        detailed = {k: torch.zeros((), device=self.device) for k in self.detailed_losses}
        for data in self.dataloader_train:
            tot, det = self.eng.train_step(self._to_device(data))
            loss_epoch += tot                                   # device-side accumulation: no .item() sync per step (train.py:312-314 syncs 12x)
            for k, v in det.items():
                detailed[k] += self.eng.detailed_weights[k] * v
            num_batches += 1
        self.log_losses(float(loss_epoch), {k: float(v) for k, v in detailed.items()}, max(num_batches, 1), '')
        self.cur_epoch += 1

    @torch.inference_mode()
    def validate(self):
        self.eng.model.eval()
        num_batches, loss_epoch = 0, 0.0
        detailed = {k: 0.0 for k in self.detailed_losses}
        for data in self.dataloader_val:
            losses = self.eng.load_data_compute_loss(self._to_device(data))
            for k, v in losses.items():
                val = self.eng.detailed_weights[k] * float(v)
                detailed[k] += val
                loss_epoch += val
            num_batches += 1
        self.log_losses(loss_epoch, detailed, max(num_batches, 1), 'val_')
        self.eng.model.train()

    def log_losses(self, loss_epoch, detailed_losses_epoch, num_batches, prefix=''):
        import json
        loss_epoch = loss_epoch / num_batches
        detailed = {k: v / num_batches for k, v in detailed_losses_epoch.items()}
        gathered_detailed, gathered_loss = [None] * self.world_size, [None] * self.world_size
        if self.parallel and self.world_size > 1:
            dist.gather_object(detailed, gathered_detailed if self.rank == 0 else None, dst=0)
            dist.gather_object(loss_epoch, gathered_loss if self.rank == 0 else None, dst=0)
        else:
            gathered_detailed[0], gathered_loss[0] = detailed, loss_epoch
        if self.rank == 0:
            rec = {prefix + "loss_total": sum(gathered_loss) / len(gathered_loss)}
            for k in detailed:
                rec[prefix + k] = sum(g[k] for g in gathered_detailed) / self.world_size
            if self.writer is not None:
                for k, v in rec.items():
                    self.writer.add_scalar(k, v, self.cur_epoch)
            with open(os.path.join(self.args.logdir, "losses.jsonl"), "a") as f:
                f.write(json.dumps(dict(epoch=self.cur_epoch, **rec)) + "\n")
            print("epoch %d %s" % (self.cur_epoch, " ".join("%s=%.4f" % kv for kv in rec.items())), flush=True)

    def save(self):
        """train.py:204-210,381-384."""
        osd = self.eng.optimizer.state_dict(self.eng.reducer.group)         # collective under ZeRO (consolidate_state_dict)
        if self.rank == 0:
            sd = self.eng.model.state_dict()
            if self.parallel:   # the reference saves the DDP-wrapped module: keys carry 'module.' (submission_agent.py:94-96 strips it)
                sd = {"module." + k: v for k, v in sd.items()}
            torch.save(sd, os.path.join(self.args.logdir, 'model_%d.pth' % self.cur_epoch))
            torch.save({k: (v.cpu() if torch.is_tensor(v) else v) for k, v in osd.items()}, os.path.join(self.args.logdir, 'optimizer_%d.pth' % self.cur_epoch))


def build_parser():
    import argparse
    p = argparse.ArgumentParser(description="MI355X-native TransFuser training (flags of team_code_transfuser/train.py:30-70)")
    p.add_argument('--id', type=str, default='transfuser')
    p.add_argument('--epochs', type=int, default=41)
    p.add_argument('--lr', type=float, default=1e-4)
    p.add_argument('--batch_size', type=int, default=12, help='per GPU; effective batch = batch_size * num_gpus (the lr is NOT scaled, as in the reference)')
    p.add_argument('--logdir', type=str, default='log')
    p.add_argument('--load_file', type=str, default=None)
    p.add_argument('--start_epoch', type=int, default=0)
    p.add_argument('--setting', type=str, default='all')
    p.add_argument('--root_dir', type=str, default='synthetic:64', help="dataset root in the reference's on-disk format, or 'synthetic:N' (N seeded samples of the dataset's shapes)")
    p.add_argument('--schedule', type=int, default=1)
    p.add_argument('--schedule_reduce_epoch_01', type=int, default=30)
    p.add_argument('--schedule_reduce_epoch_02', type=int, default=40)
    p.add_argument('--backbone', type=str, default='transFuser')
    p.add_argument('--image_architecture', type=str, default='regnety_032')
    p.add_argument('--lidar_architecture', type=str, default='regnety_032')
    p.add_argument('--use_velocity', type=int, default=0)
    p.add_argument('--n_layer', type=int, default=4)
    p.add_argument('--wp_only', type=int, default=0)
    p.add_argument('--use_target_point_image', type=int, default=1)
    p.add_argument('--use_point_pillars', type=int, default=0)
    p.add_argument('--parallel_training', type=int, default=1, help='1: launched by torchrun (RANK / LOCAL_RANK / WORLD_SIZE in the environment); 0: single process')
    p.add_argument('--val_every', type=int, default=5)
    p.add_argument('--no_bev_loss', type=int, default=0)
    p.add_argument('--sync_batch_norm', type=int, default=0)
    p.add_argument('--zero_redundancy_optimizer', type=int, default=0)
    p.add_argument('--use_disk_cache', type=int, default=0, help='1: cache the decoded samples under $SCRATCH/dataset_cache (train.py:77-90); needs a real --root_dir')
    # additions of this framework
    p.add_argument('--use_graph', type=int, default=1, help='capture the step into hipGraphs (falls back to eager where a flag requires it)')
    p.add_argument('--precision', type=str, default='fp32', choices=['fp32', 'f32x3', 'bf16'])
    p.add_argument('--height', type=int, default=160, help="RGB height of the synthetic samples (the dataset's crop is 160 x 704)")
    p.add_argument('--num_workers', type=int, default=None)
    p.add_argument('--width', type=int, default=704)
    return p


def main(argv=None):
    import datetime
    import json
    from torch.utils.data import DataLoader
    from .config import GlobalConfig
    from .data import make_datasets
    from .model import LidarCenterNet
    args = build_parser().parse_args(argv)
    args.logdir = os.path.join(args.logdir, args.id)
    parallel = bool(args.parallel_training) and "RANK" in os.environ
    cuda = torch.cuda.is_available()
    if parallel:   # train.py:94-106
        rank, local_rank, world_size = int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"]), int(os.environ["WORLD_SIZE"])
        if cuda:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend='nccl' if cuda else 'gloo', init_method='env://', world_size=world_size, rank=rank, timeout=datetime.timedelta(minutes=15))
        dist.barrier()
    else:
        rank, local_rank, world_size = 0, 0, 1
    device = torch.device('cuda:%d' % local_rank) if cuda else torch.device('cpu')
    config = GlobalConfig(root_dir=args.root_dir if os.path.isdir(args.root_dir) else '', setting=args.setting)   # train.py:113-121
    config.use_target_point_image = bool(args.use_target_point_image)
    config.n_layer = args.n_layer
    config.use_point_pillars = bool(args.use_point_pillars)
    config.backbone = args.backbone
    if bool(args.no_bev_loss):
        config.detailed_losses_weights[config.detailed_losses.index("loss_bev")] = 0.0
    model = LidarCenterNet(config, device, args.backbone, args.image_architecture, args.lidar_architecture, bool(args.use_velocity)).to(device)
    print('Total trainable parameters: ', sum(p.numel() for p in model.parameters() if p.requires_grad))
    if args.load_file is not None:      # train.py:180-183 (accepts checkpoints with or without DDP's 'module.' prefix)
        model.load_reference_checkpoint(args.load_file) if hasattr(model, "load_reference_checkpoint") else model.load_state_dict(torch.load(args.load_file, map_location=device))
    eng = Engine(model, config, lr=args.lr, use_graph=bool(args.use_graph) and cuda, wp_only=bool(args.wp_only),
                 zero_redundancy_optimizer=bool(args.zero_redundancy_optimizer), sync_batch_norm=bool(args.sync_batch_norm), precision=args.precision if cuda else None)
    if args.load_file is not None and os.path.exists(args.load_file.replace("model_", "optimizer_")):
        eng.optimizer.load_state_dict(torch.load(args.load_file.replace("model_", "optimizer_"), map_location=device))
    shared_dict = None
    if bool(args.use_disk_cache):       # train.py:77-90: decoded samples cached on the fast local storage ($SCRATCH), shared by all ranks
        if str(args.root_dir).startswith("synthetic"):
            raise ValueError("--use_disk_cache 1 caches samples decoded from a dataset directory; --root_dir %s has nothing to cache" % args.root_dir)
        import tempfile
        shared_dict = os.path.join(os.environ.get('SCRATCH') or tempfile.gettempdir(), "dataset_cache")
        print("Tmp folder for dataset cache: ", shared_dict)
    train_set, val_set = make_datasets(args.root_dir, config, height=args.height, width=args.width, shared_dict=shared_dict)
    g = torch.Generator(device='cpu')
    g.manual_seed(torch.initial_seed())
    nw = args.num_workers if args.num_workers is not None else (8 if parallel else 0)
    if parallel:    # train.py:156-161
        sampler_train = torch.utils.data.distributed.DistributedSampler(train_set, shuffle=True, num_replicas=world_size, rank=rank)
        sampler_val = torch.utils.data.distributed.DistributedSampler(val_set, shuffle=True, num_replicas=world_size, rank=rank)
        dl_train = DataLoader(train_set, sampler=sampler_train, batch_size=args.batch_size, worker_init_fn=seed_worker, generator=g, num_workers=nw, pin_memory=cuda, drop_last=bool(args.use_graph))
        dl_val = DataLoader(val_set, sampler=sampler_val, batch_size=args.batch_size, worker_init_fn=seed_worker, generator=g, num_workers=nw, pin_memory=cuda)
    else:
        sampler_train = None
        dl_train = DataLoader(train_set, shuffle=True, batch_size=args.batch_size, worker_init_fn=seed_worker, generator=g, num_workers=nw, pin_memory=cuda, drop_last=bool(args.use_graph))
        dl_val = DataLoader(val_set, shuffle=True, batch_size=args.batch_size, worker_init_fn=seed_worker, generator=g, num_workers=nw, pin_memory=cuda)
    if rank == 0:
        os.makedirs(args.logdir, exist_ok=True)
        with open(os.path.join(args.logdir, 'args.txt'), 'w') as f:      # train.py:172-175 (the agent reads it back, submission_agent.py:41-60)
            json.dump(args.__dict__, f, indent=2)
    if parallel:
        dist.barrier()
    trainer = Trainer(eng, dl_train, dl_val, args, config, device, rank, world_size, parallel, cur_epoch=args.start_epoch)
    lr = args.lr
    for epoch in range(trainer.cur_epoch, args.epochs):
        if parallel:
            sampler_train.set_epoch(epoch)        # train.py:191-193
        if epoch in (args.schedule_reduce_epoch_01, args.schedule_reduce_epoch_02) and args.schedule == 1:      # train.py:194-199
            lr *= 0.1
            print("Reduce learning rate by factor 10 to:", lr)
            eng.optimizer.set_lr(lr)
        trainer.train()
        if args.setting != 'all' and epoch % args.val_every == 0:
            trainer.validate()
        trainer.save()                            # every rank: the ZeRO state is gathered collectively, rank 0 writes
    if parallel:
        dist.destroy_process_group()
    return trainer


if __name__ == "__main__":
    main()
