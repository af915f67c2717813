"""Multi-GPU host logic: one process per GPU, users block-partitioned, collectives by torch.distributed.

Each rank owns a contiguous block of dense users.  Per step:
  1. every rank fits on the (replicated) training triples; the per-user means / norms and the
     per-rating deviations are computed for the OWNED users only (K2/K3 are row-local);
  2. the one exchange step: all-gather of the owned segments of {user mean, user norm, normalized
     deviation, preprocessed rating} so every rank can densify any user's row for the similarity
     GEMM's B operand and gather any neighbour's deviation (RCCL all-gather over xGMI);
  3. every rank builds complete top-k rows for its own users (no cross-GPU merge) and predicts the
     test ratings of its own users;
  4. all-reduce of (sum |r - p|, count) -> MAE.
The engine object only needs: fit_device, shard_tensors, shard_commit, mae_device — the GPU engine
(knncf.Engine through DeviceEngineAdapter) on the GPU box, an oracle-backed double in the CPU tests.
"""
import numpy as np


class _CudaArrayView:
    """Zero-copy view of library-owned device memory for torch (plumbing only)."""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}


def wrap_device_f64(ptr, n, device):
    import torch

    if n == 0:
        return torch.empty(0, dtype=torch.float64, device=device)
    return torch.as_tensor(_CudaArrayView(ptr, n, "<f8"), device=device)


class DeviceEngineAdapter:
    """knncf.Engine with the shard buffers exposed as torch tensors."""

    def __init__(self, engine, device):
        self.engine = engine
        self.device = device

    def fit_device(self, users, items, ratings):
        self.engine.fit_device(users, items, ratings)

    def shard_tensors(self):
        v = self.engine.shard_view()
        dev = self.device
        return {
            "user_range": (v.user_begin, v.user_end),
            "nnz_range": (v.nnz_begin, v.nnz_end),
            "user_avg": wrap_device_f64(v.d_user_avg, v.num_users, dev),
            "user_norm": wrap_device_f64(v.d_user_norm, v.num_users, dev),
            "dev": wrap_device_f64(v.d_dev, v.num_ratings, dev),
            "pre": wrap_device_f64(v.d_pre, v.num_ratings, dev),
        }

    def shard_commit(self):
        self.engine.shard_commit()

    def mae_device(self, predictor, users, items, ratings):
        return self.engine.mae_device(predictor, users, items, ratings)


def user_block(num_users, rank, world):
    """Owned dense users [lo, hi): ceil(U / world) per rank, ascending dense index."""
    per = -(-num_users // world)
    return min(per * rank, num_users), min(per * (rank + 1), num_users)


def _staged(dist, t):
    """gloo rehearsals of the multi-GPU path (several ranks sharing one GPU, or CPU tests): gloo moves host memory, so
    device tensors are staged through the host; with nccl (RCCL over xGMI) tensors travel as they are."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _all_gather_into(dist, recv, send):
    if _staged(dist, send):
        r = recv.cpu()
        dist.all_gather_into_tensor(r, send.cpu())
        recv.copy_(r)
    else:
        dist.all_gather_into_tensor(recv, send)


def _all_reduce(dist, t, op=None):
    kw = {} if op is None else {"op": op}
    if _staged(dist, t):
        c = t.cpu()
        dist.all_reduce(c, **kw)
        t.copy_(c)
    else:
        dist.all_reduce(t, **kw)


def _all_gather_segments(dist, arrays, lo, hi, ranges):
    """In-place all-gather of arrays[j][lo_r:hi_r] from every rank r.  Segments differ in length, so
    they travel in one padded all_gather_into_tensor per call (bigger, fewer collectives)."""
    import torch

    world = len(ranges)
    seg = max(h - l for l, h in ranges)
    if seg == 0:
        return
    k = len(arrays)
    send = torch.zeros(k * seg, dtype=arrays[0].dtype, device=arrays[0].device)
    for j, a in enumerate(arrays):
        send[j * seg: j * seg + (hi - lo)] = a[lo:hi]
    recv = torch.empty(world * k * seg, dtype=send.dtype, device=send.device)
    _all_gather_into(dist, recv, send)
    for r, (l, h) in enumerate(ranges):
        if (l, h) == (lo, hi):
            continue
        base = r * k * seg
        for j, a in enumerate(arrays):
            a[l:h] = recv[base + j * seg: base + j * seg + (h - l)]


class ShardFitError(RuntimeError):
    """Another rank's fit failed; every rank raises so that none blocks in the exchange."""

    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


class ShardedKnn:
    def __init__(self, engine, dist=None, rank=0, world=1):
        self.engine, self.dist, self.rank, self.world = engine, dist, rank, world

    def fit(self, users, items, ratings):
        """fit + the one exchange + commit.  The fit STATUS is collective: a rank whose part of the fit fails (a
        non-finite deviation among ITS users, a device error, out of memory) must not leave the others waiting in the
        all-gather, so every rank first learns whether all of them succeeded and all of them raise otherwise."""
        err = None
        try:
            self.engine.fit_device(users, items, ratings)
        except Exception as e:  # reported collectively below
            err = e
        if self.world > 1:
            self._raise_together(err, users.device)
            self.exchange()
        elif err is not None:
            raise err
        self.engine.shard_commit()

    def _raise_together(self, err, device):
        import torch

        code = 0 if err is None else int(getattr(err, "status", -1)) or -1
        flag = torch.tensor([code], dtype=torch.int64, device=device)
        _all_reduce(self.dist, flag, op=self.dist.ReduceOp.MIN)   # status codes are negative: MIN = "the worst"
        worst = int(flag.item())
        if err is not None:
            raise err
        if worst != 0:
            raise ShardFitError(worst, f"fit failed on another rank (status {worst}); this rank's part was fine")

    def exchange(self):
        import torch

        t = self.engine.shard_tensors()
        ulo, uhi = t["user_range"]
        nlo, nhi = t["nnz_range"]
        dev = t["user_avg"].device
        mine = torch.tensor([ulo, uhi, nlo, nhi], dtype=torch.int64, device=dev)
        allr = torch.empty(4 * self.world, dtype=torch.int64, device=dev)
        _all_gather_into(self.dist, allr, mine)
        allr = allr.cpu().view(self.world, 4).tolist()
        _all_gather_segments(self.dist, [t["user_avg"], t["user_norm"]], ulo, uhi, [(a, b) for a, b, _, _ in allr])
        _all_gather_segments(self.dist, [t["dev"], t["pre"]], nlo, nhi, [(c, d) for _, _, c, d in allr])
        # the engine works on its own (non-blocking) HIP stream: the gathered segments must have landed before
        # shard_commit launches the kernels that read them
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()

    def mae(self, predictor, users, items, ratings):
        """All-reduced MAE of the whole test set; every rank receives the same value."""
        import torch

        s, c = self.engine.mae_device(predictor, users, items, ratings)
        if self.world > 1:
            buf = torch.tensor([s, float(c)], dtype=torch.float64, device=users.device)
            _all_reduce(self.dist, buf)
            s, c = float(buf[0].item()), int(round(buf[1].item()))
        return (s / c if c else float("nan")), c
