"""Data-parallel plumbing for the HSTU hot path: one process per GPU, the batch of users is
sharded across ranks (every op on the path is per-user, so the forward needs no
collective), and the layer-parameter gradients are summed with ONE bucketed all-reduce per
step on RCCL over xGMI (``backend="nccl"`` under PyTorch-ROCm; ``gloo`` in CPU tests).

Mirrors the role of DistributedSampler + DDP in the reference trainer
(research/trainer/data_loader.py:39-46, research/trainer/train.py:73-78,269) without
wrapping the model: the attention-only benchmark has no parameters and therefore no
collective at all; the layer benchmark calls ``GradientAllReducer.reduce()`` after backward.
"""

from __future__ import annotations

import os
from typing import Iterable, List, Optional, Sequence, Tuple, Union

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, single_rank_group: bool = False) -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from torchrun-style env; initialises the process group
    when WORLD_SIZE > 1 -- or, with ``single_rank_group``, also for a lone rank (a one-rank RCCL
    communicator: the only way to put the library and the reducer's stream discipline on the
    hardware of a one-GPU box).  MASTER_ADDR defaults to 127.0.0.1 (container hostnames may not
    resolve)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or single_rank_group) and not dist.is_initialized():
        # Hosts whose driver only supports dmabuf IPC: without this RCCL's buffer exchange fails in hipIpcGetMemHandle.  The HSA
        # runtime reads it when the process first touches the GPU -- so it only helps when nothing has touched it yet.
        if "HSA_ENABLE_IPC_MODE_LEGACY" not in os.environ:
            if torch.cuda.is_available() and torch.cuda.is_initialized():
                import warnings

                warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY is not set and the GPU runtime is already initialised: on dmabuf-only hosts "
                              "RCCL will fail with 'hipIpcGetMemHandle: invalid argument'; export HSA_ENABLE_IPC_MODE_LEGACY=0 before "
                              "the first torch.cuda call")
            else:
                os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            # HSTU_DIST_BACKEND=gloo: rehearse the multi-rank flow where there are fewer GPUs than ranks (RCCL refuses two
            # ranks on one device; gloo moves the few control tensors through the host)
            backend = os.environ.get("HSTU_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            ndev = torch.cuda.device_count()
            if local_rank >= ndev:
                # two ranks on one device deadlock or fail inside RCCL at the first collective; say so here instead
                raise RuntimeError(
                    f"rank {rank} (LOCAL_RANK {local_rank}) has no GPU of its own: this node exposes {ndev} device(s) for "
                    f"{world} ranks.  RCCL needs one device per rank; set HSTU_DIST_BACKEND=gloo to rehearse the "
                    f"multi-rank flow on fewer GPUs.")
            torch.cuda.set_device(local_rank)      # RCCL binds a communicator to the device current at first use
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def describe_ranks(device: Union[torch.device, str] = "cpu") -> dict:
    """Who is in the job: backend, library version and every rank's (host, device index, device name, PCI bus id),
    gathered over the process group -- what NCCL_DEBUG=VERSION / INFO would print, as data for the bench line."""
    import socket

    me = {"rank": dist.get_rank() if dist.is_initialized() else 0, "host": socket.gethostname(), "pid": os.getpid()}
    if torch.cuda.is_available() and str(device).startswith("cuda"):
        i = torch.cuda.current_device()
        pr = torch.cuda.get_device_properties(i)
        me.update(device=i, name=pr.name, bus_id=getattr(pr, "pci_bus_id", None), gcn_arch=getattr(pr, "gcnArchName", None))
    info = {"backend": dist.get_backend() if dist.is_initialized() else None, "ranks": [me]}
    if dist.is_initialized() and dist.get_world_size() > 1:
        allr: List[Optional[dict]] = [None] * dist.get_world_size()
        dist.all_gather_object(allr, me)
        info["ranks"] = allr
    if info["backend"] == "nccl":
        try:
            info["rccl_version"] = ".".join(str(x) for x in torch.cuda.nccl.version())
        except Exception as e:  # pragma: no cover
            info["rccl_version"] = repr(e)
        info["env"] = {k: os.environ[k] for k in ("NCCL_DEBUG", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_SOCKET_IFNAME", "RCCL_MSCCL_ENABLE")
                       if k in os.environ}
    devs = [(r.get("host"), r.get("device")) for r in info["ranks"] if r and "device" in r]
    info["one_device_per_rank"] = len(set(devs)) == len(devs)
    return info


def shard_users(lengths: torch.Tensor, rank: int, world_size: int, balance: str = "count") -> torch.Tensor:
    """Indices of the users owned by ``rank``.

    ``count``: contiguous equal-count slices [g*B/G, (g+1)*B/G).
    ``work`` : greedy longest-processing-time assignment on L^2 (attention work), for
               long-tailed length distributions; deterministic, identical on every rank.
    """
    B = lengths.numel()
    if world_size == 1:
        return torch.arange(B)
    if balance == "count":
        lo, hi = rank * B // world_size, (rank + 1) * B // world_size
        return torch.arange(lo, hi)
    if balance != "work":
        raise ValueError(f"unknown balance mode {balance}")
    work = lengths.to(torch.float64).cpu() ** 2
    order = torch.argsort(work, descending=True, stable=True).tolist()
    load = [0.0] * world_size
    mine: List[int] = []
    for u in order:
        g = min(range(world_size), key=lambda r: (load[r], r))
        load[g] += float(work[u])
        if g == rank:
            mine.append(u)
    return torch.tensor(sorted(mine), dtype=torch.int64)


def local_offsets(lengths: torch.Tensor) -> torch.Tensor:
    """Re-based seq_offsets of a shard ([0, cumsum(lengths)], int64)."""
    off = torch.zeros(lengths.numel() + 1, dtype=torch.int64, device=lengths.device)
    off[1:] = torch.cumsum(lengths.to(torch.int64), 0)
    return off


class GradientAllReducer:
    """Flat-bucket gradient all-reduce (sum or mean).

    xGMI is point-to-point (7 links x ~153 GB/s per GPU) and the in-scope parameters are
    small (3 STU layers at D=512: 5.5 M params = 22 MB fp32), so the whole
    gradient fits ONE bucket: a single large collective keeps every link busy and pays the
    launch latency once.  Larger models split at ``bucket_bytes``.
    """

    # overlap mode: every bucket carries ONE extra element behind its gradients, set to a small integer that names the bucket
    # before its all-reduce is issued.  Collectives pair up across ranks by issue order, so if the ranks ever disagree on that
    # order (a bucket launched from a hook on one rank and late on another) buffers of DIFFERENT buckets are summed -- and with
    # equally sized per-layer buckets nothing fails.  The tag makes that visible in the data the mispaired collective itself
    # returns: every rank that took part in a wrong pairing reads a sum that is not world x its own tag.  No extra collective,
    # nothing that could itself mispair; reading the tags is one small device-to-host copy, done on the first ``check_first``
    # calls of reduce() and every ``check_every``-th after that.
    TAG_MOD = 7      # tags 1..7: world x tag <= 56 stays exact in bf16 buckets up to 8 ranks (mean: <= 7 everywhere)

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_bytes: int = 64 << 20, average: bool = True,
                 buckets: Optional[Sequence[Iterable[torch.nn.Parameter]]] = None, overlap: bool = False,
                 check_every: int = 50, check_first: int = 3, single_rank_collectives: bool = False):
        """``buckets``: explicit parameter groups (e.g. one per layer) instead of the byte-size split of ``params``.
        ``overlap``: launch a bucket's all-reduce from inside backward, as soon as its last gradient has been
        accumulated (hooks); ``reduce()`` then only waits.  The reference's DDP does the same (train.py:269).
        ``single_rank_collectives``: issue the collectives also in a one-rank group (tests / bench on a one-GPU box:
        the hooks, the staging copy, the asynchronous all-reduce on the communicator's stream and the hand-back into
        ``p.grad`` then run exactly as they do with 8 ranks; default: a lone rank skips them)."""
        self.average = average
        self.check_every = int(check_every)     # overlap mode: the bucket tags are read back on every N-th reduce() (0: only the first calls)
        self.check_first = int(check_first)
        self.single_rank_collectives = bool(single_rank_collectives)
        self.buckets: List[List[torch.nn.Parameter]] = []
        if buckets is not None:
            # a bucket is ONE flat buffer: parameters of different dtypes (or devices) of a group get buckets of their own,
            # in first-appearance order (as the byte-size split below does)
            for grp in buckets:
                by_kind = {}
                for p in grp:
                    if p.requires_grad:
                        by_kind.setdefault((p.dtype, p.device), []).append(p)
                self.buckets.extend(by_kind.values())
            self.params = [p for b in self.buckets for p in b]
        else:
            self.params = [p for p in params if p.requires_grad]
            cur: List[torch.nn.Parameter] = []
            cur_bytes = 0
            for p in self.params:
                nbytes = p.numel() * p.element_size()
                if cur and (cur_bytes + nbytes > bucket_bytes or cur[0].dtype != p.dtype):
                    self.buckets.append(cur)
                    cur, cur_bytes = [], 0
                cur.append(p)
                cur_bytes += nbytes
            if cur:
                self.buckets.append(cur)
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self.overlap = overlap
        self._sync = True
        self._pending: List[int] = [0] * len(self.buckets)
        self._work: List[Optional[object]] = [None] * len(self.buckets)
        self._hooks = []
        if overlap:
            self._slot = {}
            for bi, bucket in enumerate(self.buckets):
                o = 0
                for p in bucket:
                    self._slot[id(p)] = (bi, o)
                    o += p.numel()
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
            self._reset()

    # ---- overlapped mode: gradients land in the bucket's flat buffer as backward produces them; the bucket's collective
    # starts when its last gradient is in (for a stack of layers: layer l's bucket reduces while layer l-1 runs backward)
    def _reset(self) -> None:
        self._pending = [len(b) for b in self.buckets]
        self._work = [None] * len(self.buckets)
        self._staged = [[] for _ in self.buckets]

    def _bucket_flat(self, bi: int, like: torch.Tensor) -> torch.Tensor:
        """the bucket's flat buffer: allocated once, sized and typed from its parameters (a gradient of another dtype is
        refused: re-allocating would orphan the views already handed out as p.grad)"""
        bucket = self.buckets[bi]
        if like.dtype != bucket[0].dtype or like.device != bucket[0].device:
            raise RuntimeError(f"GradientAllReducer: gradient of dtype {like.dtype} on {like.device} for a bucket of "
                               f"{bucket[0].dtype} parameters on {bucket[0].device}")
        flat = self._flat[bi]
        if flat is None:
            n = sum(p.numel() for p in bucket)
            flat = torch.empty(n + 1, dtype=bucket[0].dtype, device=bucket[0].device)      # + the bucket's tag
            self._flat[bi] = flat
        return flat

    def _collectives_on(self) -> bool:
        return dist.is_initialized() and (dist.get_world_size() > 1 or self.single_rank_collectives)

    def _tag(self, bi: int) -> float:
        return float(bi % self.TAG_MOD + 1)

    def _check_tags(self, reduced: List[int], world: int) -> None:
        """the tags of the buckets all-reduced in this call, read back in one copy: world x tag (tag after the mean)"""
        if not reduced:
            return
        got = torch.stack([self._flat[bi][-1].float() for bi in reduced]).cpu().tolist()
        bad = [(bi, g) for bi, g in zip(reduced, got) if g != (self._tag(bi) if self.average else self._tag(bi) * world)]
        if bad:
            raise RuntimeError(
                "GradientAllReducer(overlap=True): bucket tag mismatch after the all-reduce for bucket(s) "
                f"{[bi for bi, _ in bad]} (read {[g for _, g in bad]}): the ranks issued their bucket collectives in different "
                "orders, so gradients of different buckets were summed.  The set of parameters that receive a gradient in a "
                "step must be the same on every rank (or use overlap=False).")

    def no_sync(self):
        """context manager for gradient accumulation: backward passes inside it only accumulate locally (into the bucket
        buffers: p.grad becomes a view of its bucket and later passes add in place); the first backward OUTSIDE launches the
        collectives on the accumulated sums (the role of DDP.no_sync())."""
        import contextlib

        @contextlib.contextmanager
        def ctx():
            old, self._sync = self._sync, False
            try:
                yield
            finally:
                self._sync = old

        return ctx()

    def _on_grad(self, p: torch.nn.Parameter) -> None:
        bi, o = self._slot[id(p)]
        if self._work[bi] is not None:
            raise RuntimeError("GradientAllReducer: a gradient arrived for a bucket whose all-reduce is already in flight -- call "
                               "reduce() after every backward, or wrap the extra backward passes in no_sync()")
        flat = self._bucket_flat(bi, p.grad)
        view = flat[o : o + p.numel()].view_as(p)
        if p.grad.data_ptr() != view.data_ptr():
            self._staged[bi].append((p, view))       # moved into the bucket when its last gradient is in: ONE multi-tensor copy
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            staged = self._staged[bi]
            if staged:
                # (a hipMemcpyDtoD per parameter costs ~15 us each on the compute stream, 24 per step of a 3-layer stack;
                # _foreach_copy_ is one kernel launch per bucket)
                torch._foreach_copy_([v for _, v in staged], [q.grad for q, _ in staged])
                for q, v in staged:
                    q.grad = v         # the reduced values appear in p.grad without a copy back
                self._staged[bi] = []
            if not self._sync:
                self._pending[bi] = len(self.buckets[bi])        # accumulate only: count the next backward from the top
            elif self._collectives_on():
                flat[-1] = self._tag(bi)
                self._work[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []

    def reduce(self) -> None:
        if self.overlap:
            world = dist.get_world_size() if dist.is_initialized() else 1
            # Collectives pair up across ranks by ISSUE ORDER.  The hooks launch a bucket when its last gradient arrives, i.e. in
            # backward order, identical on every rank as long as every parameter gets a gradient everywhere.  A bucket that is
            # incomplete on this rank (parameters unused in this step) is reduced here -- zeros for the missing gradients --
            # in bucket order AFTER the hook-launched ones; a rank on which the same bucket was complete launched it from a
            # hook already, and the two would pair different buffers.  PRECONDITION: the set of parameters that receive a
            # gradient in a step is the same on every rank -- the model's parameter usage may not depend on a rank's data
            # (otherwise: overlap=False).  A violation cannot be prevented here (by the time it could be seen the mispaired
            # collectives have been issued) but it is DETECTED: see the bucket tags above.
            late = [bi for bi, w in enumerate(self._work) if w is None and self._pending[bi] != 0]
            coll = self._collectives_on()
            reduced = []
            for bi, work in enumerate(self._work):
                if work is not None:
                    work.wait()
                    if self.average:
                        self._flat[bi].div_(world)
                    reduced.append(bi)
            self._calls = getattr(self, "_calls", 0) + 1
            if late and coll and self._sync and not getattr(self, "_warned_late", False):
                import warnings

                self._warned_late = True
                warnings.warn(f"GradientAllReducer(overlap=True): bucket(s) {late} had parameters without a gradient in this step; "
                              "they are reduced after the hook-launched ones, which is only correct when every rank sees the same set")
            for bi in late:
                staged = self._staged[bi]
                if staged:
                    flat = self._bucket_flat(bi, staged[0][0].grad)
                    torch._foreach_copy_([v for _, v in staged], [q.grad for q, _ in staged])
                    for q, v in staged:
                        q.grad = v
                    self._staged[bi] = []
                if not coll or not self._sync:
                    continue
                flat = self._bucket_flat(bi, self.buckets[bi][0])
                for p in self.buckets[bi]:
                    if p.grad is None:
                        b2, o = self._slot[id(p)]
                        flat[o : o + p.numel()].zero_()
                        p.grad = flat[o : o + p.numel()].view_as(p)
                flat[-1] = self._tag(bi)
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)
                if self.average:
                    flat.div_(world)
                reduced.append(bi)
            if coll and self._sync and (self._calls <= self.check_first or (self.check_every > 0 and self._calls % self.check_every == 0)):
                self._check_tags(reduced, world)
            self._reset()
            return
        if not self._collectives_on():
            return
        world = dist.get_world_size()
        handles = []
        for i, bucket in enumerate(self.buckets):
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in bucket]
            n = sum(g.numel() for g in grads)
            flat = self._flat[i]
            if flat is None or flat.numel() != n or flat.device != grads[0].device:
                flat = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
                self._flat[i] = flat
            torch.cat([g.reshape(-1) for g in grads], out=flat)
            handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        for work, flat, bucket in handles:
            work.wait()
            if self.average:
                flat.div_(world)
            o = 0
            for p in bucket:
                n = p.numel()
                if p.grad is None:
                    p.grad = flat[o : o + n].view_as(p).clone()
                else:
                    p.grad.copy_(flat[o : o + n].view_as(p))
                o += n


def max_over_ranks(value: float, device: Union[torch.device, str] = "cpu") -> float:
    """MAX of a python float over all ranks (bench timing contract)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device: Union[torch.device, str] = "cpu") -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
