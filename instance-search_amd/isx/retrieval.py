"""Row-sharded gallery search across the GPUs of one node (BASELINE config 5).

The reference is single-device (one torch.mm, test/classif_finetune_test.py:82); this is the
MI355X-native extension the north star asks for.  One process per GPU; rank p holds gallery
rows [lo_p, hi_p) (contiguous split).  A search is
    1. local fused cosine top-k on the shard with idx_base = lo_p     (libisx isx_cosine_topk_fast:
       fp16-MFMA filter + exact fp32 re-scoring, bit-identical to isx_cosine_topk; the fp16 image of
       the shard is built once and cached)
    2. ONE all-gather of the (M,k) fp32 scores and (M,k) int64 global indices over RCCL/xGMI
       (12 B per entry: 12 MB per rank at M = 10k, k = 100 -- tiny next to the GEMM): `exchange_topk`, which on GPU tensors
       in an RCCL process group is the C ABI's `isx_shard_topk_allgather` (one grouped launch on the caller's stream)
    3. canonical merge of the P*k candidates per query                  (libisx isx_topk_merge)
The canonical comparator (score desc, GLOBAL index asc) makes the result independent of P and
of the gather order: sharded == unsharded bit for bit (tests/test_distributed.py).

`ReplicatedGallery` is the alternative for galleries that fit one GPU: whole gallery on every rank, queries split.

CPU tensors (gloo, used by the world_size-2 CPU tests and the reference's --device=-1 mode) go
through the same driver with torch doing the local top-k / merge arithmetic.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous split: the first n_rows % world_size ranks hold one extra row."""
    q, r = divmod(n_rows, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _canonical_topk_cpu(scores, idx, k):
    """k best (score desc, idx asc) per row of candidate lists; idx < 0 marks padding."""
    s = scores.clone()
    s[idx < 0] = float('-inf')
    big = torch.iinfo(torch.int64).max
    order = torch.where(idx < 0, torch.full_like(idx, big), idx).argsort(dim=1, stable=True)      # by index asc
    s1, i1 = s.gather(1, order), idx.gather(1, order)
    order2 = s1.sort(dim=1, descending=True, stable=True).indices                                  # then score desc (stable)
    return s1.gather(1, order2)[:, :k], i1.gather(1, order2)[:, :k]


def local_topk(Q, G, k, idx_base=0, ws=None, gallery_f16=None):
    """Canonical top-k of Q @ G.T on one shard, global indices.  gallery_f16: the cached
    ops.gallery_to_f16(G) pair selects the filter + exact re-scoring path (same result)."""
    M = Q.size(0)
    if Q.is_cuda:
        from . import ops
        if gallery_f16 is not None:
            return ops.cosine_topk_fast(Q, G, k, idx_base=idx_base, gallery_f16=gallery_f16, ws=ws)
        return ops.cosine_topk(Q, G, k, idx_base=idx_base, ws=ws)
    sim = Q @ G.t()
    n = G.size(0)
    order = sim.sort(dim=1, descending=True, stable=True)
    s, i = order.values[:, :k], order.indices[:, :k] + idx_base
    if n < k:
        s = torch.cat([s, sim.new_full((M, k - n), float('-inf'))], 1)
        i = torch.cat([i, i.new_full((M, k - n), -1)], 1)
    return s.contiguous(), i.contiguous()


def merge_topk(scores, idx):
    """(P,M,k) per-shard lists -> (M,k) global list."""
    P, M, k = scores.shape
    if scores.is_cuda:
        from . import ops
        return ops.topk_merge(scores, idx)
    return _canonical_topk_cpu(scores.permute(1, 0, 2).reshape(M, P * k), idx.permute(1, 0, 2).reshape(M, P * k), k)


class NativeComm(object):
    """An RCCL communicator owned by libisx (csrc/comm.cpp): the unique id is created on rank 0 and
    shipped through the existing torch.distributed group (control plane only); the data-path
    all-gather then runs as ONE grouped ncclAllGather pair issued by the library itself."""

    def __init__(self, group=None):
        from . import ops
        self.nranks = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        box = [ops.comm_unique_id() if self.rank == 0 else None]
        if self.nranks > 1:
            dist.broadcast_object_list(box, src=0, group=group)
        self.handle = ops.comm_init_rank(self.nranks, self.rank, box[0])

    def allgather_topk(self, s, i):
        from . import ops
        return ops.shard_topk_allgather(self.handle, self.nranks, s, i)

    def allgather_rows(self, rows):
        from . import ops
        return ops.comm_allgather_rows(self.handle, self.nranks, rows)

    def close(self):
        if self.handle:
            from . import ops
            ops.comm_destroy(self.handle)
            self.handle = None


_NATIVE_COMMS = {}            # torch.distributed group -> the libisx communicator opened over it (one per process and group)


_NATIVE_OFF = set()           # groups whose communicator could not be opened on EVERY rank: they exchange over torch.distributed


def native_comm_for(group=None):
    """The libisx RCCL communicator of `group`, opened on first use (a collective: every rank of the group gets here together, at its
    first exchange) and kept for the life of the process.  Opening can fail on SOME ranks only (after the unique-id broadcast, say): the ranks
    agree on the outcome with one all-reduce(MIN) of a success flag over the torch group before any of them takes a path -- otherwise the ones
    that succeeded would wait in a native all-gather for ranks that went to torch.distributed (round-4 ADVICE).  Returns None when the group
    stays on torch.distributed."""
    if group in _NATIVE_OFF:
        return None
    nc = _NATIVE_COMMS.get(group)
    if nc is None:
        err = None
        try:
            nc = NativeComm(group)
        except Exception as e:                               # librccl not loadable / communicator refused
            err = e
        ok = torch.tensor([0 if nc is None else 1], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            import sys
            if nc is not None:
                nc.close()
            print("isx.retrieval: libisx RCCL communicator unavailable on %s (%s); exchanging over torch.distributed"
                  % ("this rank" if err is not None else "another rank", "%s: %s" % (type(err).__name__, err) if err is not None else "agreed by all-reduce"),
                  file=sys.stderr)
            _NATIVE_OFF.add(group)
            return None
        _NATIVE_COMMS[group] = nc
    return nc


def close_native_comms():
    for nc in _NATIVE_COMMS.values():
        nc.close()
    _NATIVE_COMMS.clear()


def exchange_backend(group=None, cuda=True):
    """Which implementation exchange_topk takes for tensors on (cuda ? the GPU : the host) in this process group."""
    import os
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return "none (one rank)"
    if cuda and dist.get_backend(group) == "nccl" and os.environ.get("ISX_NATIVE_COMM", "1") != "0" and group not in _NATIVE_OFF:
        return "isx_shard_topk_allgather (libisx: one grouped ncclAllGather pair over RCCL)"
    return "torch.distributed all_gather_into_tensor x 2 (%s)" % dist.get_backend(group)


def exchange_topk(s, i, group=None, native_comm=None):
    """THE exchange step of the sharded search: per-shard (M,k) lists of every rank -> (P,M,k) rank-major, on every rank.
    GPU tensors in an RCCL ("nccl") process group go through the C ABI's `isx_shard_topk_allgather` -- scores and indices as ONE grouped
    RCCL launch on the caller's stream; CPU tensors / gloo groups (the world_size-2 CPU tests, ISX_BENCH_ONE_DEVICE) through
    torch.distributed.  ISX_NATIVE_COMM=0 forces the torch.distributed path (A/B)."""
    if native_comm is None and (not dist.is_initialized() or dist.get_world_size(group) == 1):
        return s[None], i[None]
    if native_comm is None and s.is_cuda and exchange_backend(group).startswith("isx_"):
        native_comm = native_comm_for(group)                 # None: the ranks agreed to stay on torch.distributed (module state, not os.environ)
    if native_comm is not None and s.is_cuda:
        if native_comm.nranks == 1:
            return s[None], i[None]
        return native_comm.allgather_topk(s.contiguous(), i.contiguous())
    P = dist.get_world_size(group)
    all_s = torch.empty((P,) + tuple(s.shape), dtype=s.dtype, device=s.device)
    all_i = torch.empty((P,) + tuple(i.shape), dtype=i.dtype, device=i.device)
    # concatenated (P*M, k) views: the layout both RCCL and gloo accept for an all-gather
    dist.all_gather_into_tensor(all_s.view(-1, s.size(1)), s.contiguous(), group=group)
    dist.all_gather_into_tensor(all_i.view(-1, i.size(1)), i.contiguous(), group=group)
    return all_s, all_i


class ShardedGallery(object):
    """This rank's slice of a row-sharded descriptor gallery."""

    def __init__(self, shard, idx_base, group=None, native_comm=None, fast=True):
        self.shard = shard.contiguous()
        self.idx_base = int(idx_base)
        self.group = group
        self.native_comm = native_comm        # an explicit NativeComm (default: the group's own, opened on first use -- exchange_topk)
        self.fast = bool(fast) and self.shard.is_cuda
        self._ws = None
        self._f16 = None                      # (Gh, gstats), built on first search
        # fallback watch: rows of the previous fast search that needed the exact fp32 pass (read back asynchronously).
        # Data with dense clusters of near-equal scores makes the filter pay without saving anything: above
        # FALLBACK_LIMIT of the rows the gallery switches itself to the all-fp32 search (same results).
        self._fb_host = None
        self._fb_event = None
        self._fb_rows = 0

    @classmethod
    def from_full(cls, gallery, group=None):
        """Slice a replicated gallery tensor by rank (tests / small galleries)."""
        ws = dist.get_world_size(group) if dist.is_initialized() else 1
        rk = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(gallery.size(0), ws, rk)
        return cls(gallery[lo:hi], lo, group)

    @classmethod
    def from_slab(cls, path, device, group=None, fast=True):
        """This rank's contiguous row range of a descriptor slab file (isx/slab.py), read straight into its device: every rank maps only its own
        1/P of the file (mmap -> pinned staging -> HBM) -- the gallery is extracted once and searched by any number of ranks, where the reference
        re-extracts it on every run (test/classif_finetune_test.py:80-81)."""
        from . import slab
        ws = dist.get_world_size(group) if dist.is_initialized() else 1
        rk = dist.get_rank(group) if dist.is_initialized() else 0
        lo, hi = shard_bounds(slab.slab_info(path)["rows"], ws, rk)
        shard, _ = slab.load_slab(path, device, rows=(lo, hi))
        return cls(shard, lo, group, fast=fast)

    def _workspace(self, M, k):
        if not self.shard.is_cuda:
            return None
        from . import ops
        if self.fast:
            need = ops.cosine_topk_fast_workspace(M, self.shard.size(0), self.shard.size(1), k, True)
        else:
            need = ops.cosine_topk_workspace(M, self.shard.size(0), self.shard.size(1), k)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty((need,), dtype=torch.uint8, device=self.shard.device)
        return self._ws

    FALLBACK_LIMIT = 0.25

    def _check_fallback(self):
        if self._fb_event is not None and self._fb_event.query():
            if self._fb_rows and int(self._fb_host[0]) > self.FALLBACK_LIMIT * self._fb_rows:
                self.fast = False                      # this data defeats the filter: stay on the fp32 search
            self._fb_event = None

    def local_search(self, Q, k):
        """This rank's shard only (no collective): canonical top-k with global indices."""
        if self.fast:
            self._check_fallback()
        if self.fast and self._f16 is None and self.shard.size(0) > 0:
            from . import ops
            self._f16 = ops.gallery_to_f16(self.shard)
        ws = self._workspace(Q.size(0), k)
        out = local_topk(Q, self.shard, k, self.idx_base, ws, self._f16 if self.fast else None)
        if self.fast and self._f16 is not None and self._fb_event is None:
            from . import ops
            cnt = ops.cosine_topk_fast_fallback_counter(ws, Q.size(0), self.shard.size(0), self.shard.size(1), k, True)
            if cnt is not None:
                if self._fb_host is None:
                    self._fb_host = torch.zeros(1, dtype=torch.int32).pin_memory()
                self._fb_host.copy_(cnt, non_blocking=True)
                self._fb_rows = Q.size(0)
                self._fb_event = torch.cuda.Event()
                self._fb_event.record(torch.cuda.current_stream(self.shard.device))
        return out

    def average_precisions(self, Q, qlab, glab_local, kth=1, budget_bytes=None):
        """float64 average precision of every (replicated) query against the WHOLE sharded gallery (reference utils/metrics.py:25-45 on one full
        score row): the ranks of the positives are counts that add over the shards (include/isx.h isx_ap_shard_*; one all-gather of the
        positives' keys, one all-reduce of their rank histograms) -- the bits of the unsharded evaluation, on every rank.
        qlab: (M,) int32 query labels, glab_local: (rows of this shard,) int32."""
        from utils.metrics import sharded_average_precisions
        return sharded_average_precisions(Q, self.shard, self.idx_base, qlab, glab_local, kth, self.group, budget_bytes)

    def search(self, Q, k):
        """Global canonical top-k for the (replicated) query block Q: (scores (M,k), idx (M,k))."""
        s, i = self.local_search(Q, k)
        all_s, all_i = exchange_topk(s, i, self.group, self.native_comm)
        if all_s.size(0) == 1:
            return s, i
        return merge_topk(all_s, all_i)


class ReplicatedGallery(object):
    """The other way to use N GPUs, for galleries that fit one MI355X (a 1 M x 2048 gallery is 8.2 GB fp32 + 4.1 GB for the
    cached fp16 image, of 288 GB): every rank holds the WHOLE gallery and searches its own slice of the query block;
    the per-rank lists are concatenated by one all-gather -- no merge step, nothing to tie-break across ranks, and the
    per-query costs (bootstrap, re-scoring) are divided by N as well.  Measured per GPU at 8 ranks' share of config 5
    (1250 queries x 1 M rows): 7.6 ms, against 8.4 ms + all-gather + merge for a 125 k-row shard and 10 k queries.
    Result: identical to ShardedGallery.search and to the unsharded search (rows are independent)."""

    def __init__(self, gallery, group=None, fast=True):
        self._local = ShardedGallery(gallery, 0, group=None, fast=fast)       # the local search machinery, no collective
        self.group = group

    def search(self, Q, k):
        """Q: the replicated query block (M, D).  Returns the (M, k) lists on every rank."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return self._local.local_search(Q, k)
        P, rank = dist.get_world_size(self.group), dist.get_rank(self.group)
        M = Q.size(0)
        lo, hi = shard_bounds(M, P, rank)
        per = (M + P - 1) // P                                  # rows per rank, padded to equal blocks for the all-gather
        s = torch.full((per, k), float('-inf'), dtype=torch.float32, device=Q.device)
        i = torch.full((per, k), -1, dtype=torch.int64, device=Q.device)
        if hi > lo:
            ls, li = self._local.local_search(Q[lo:hi].contiguous(), k)
            s[:hi - lo], i[:hi - lo] = ls, li
        all_s = torch.empty((P * per, k), dtype=s.dtype, device=s.device)
        all_i = torch.empty((P * per, k), dtype=i.dtype, device=i.device)
        dist.all_gather_into_tensor(all_s, s, group=self.group)
        dist.all_gather_into_tensor(all_i, i, group=self.group)
        keep = torch.cat([torch.arange(p * per, p * per + (shard_bounds(M, P, p)[1] - shard_bounds(M, P, p)[0]), device=Q.device)
                          for p in range(P)])
        return all_s.index_select(0, keep), all_i.index_select(0, keep)


def gather_queries(q_local, group=None):
    """Data-parallel extraction -> replicated query block: all-gather the per-rank descriptor
    rows (rank order), equal row counts per rank.  GPU rows in an RCCL group travel on the SAME libisx communicator and stream as the result
    exchange of the search that follows (isx_comm_allgather_rows, then isx_shard_topk_allgather): one communicator, program order on every
    rank.  When the ranks agreed to stay on torch.distributed (native_comm_for -> None), gloo groups and CPU tensors, both all-gathers go
    through torch.distributed -- again one communicator."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return q_local
    if q_local.is_cuda and q_local.dtype == torch.float32 and exchange_backend(group).startswith("isx_"):
        nc = native_comm_for(group)
        if nc is not None:
            return nc.allgather_rows(q_local.contiguous())
    P = dist.get_world_size(group)
    out = torch.empty((P * q_local.size(0), q_local.size(1)), dtype=q_local.dtype, device=q_local.device)
    dist.all_gather_into_tensor(out, q_local.contiguous(), group=group)
    return out
