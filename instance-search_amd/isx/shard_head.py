"""The descriptor head's Linear(100352 -> D) SHARDED BY OUTPUT FEATURES across the ranks of a data-parallel training run (BASELINE configs[3];
reference model/siamese.py:104-114: ONE 822 MB weight, replicated on every device by a plain data-parallel port).

Round 4 kept the head replicated: the ranks all-gathered the (x, dy) rows and EVERY rank formed the whole 822 MB weight gradient and updated the
whole weight -- at P = 8 a third of a rank's step.  Here rank r owns the output features [r D / P, (r + 1) D / P): its rows of W, their momentum
and their update.  One optimizer step:

  forward    all-gather the head inputs x (R x K rows of every rank, rank order = micro-batch order)      -- the rows round 4 already exchanged
             y[:, own] = x_all . W_own^T + b_own   for ALL rows, then all-gather the column slices        -- (R, D) floats: 1.5 MB
  backward   all-gather dy (R x D); rows (x_all, dy_all[:, own]) -> this rank's part of dW, formed and applied by ONE kernel (isx_head_sgd_step)
             input gradient: every rank computes the chains of ITS feature groups for all rows (isx_head_linear_dgrad_parts), an all-to-all
             hands every row's pieces to the rank that owns the row, which adds the G = 8 group sums in group order
  sync       the updated rows of W are all-gathered only when somebody needs the whole weight (the epoch's embedding pass, an evaluation, a
             checkpoint) -- not per step.

Bit-identity with one process: an output y[m][n] and a gradient dW[n][k] are computed whole by one rank with the kernels the single process
uses; dx is DEFINED (csrc/head.hip isx_head_linear_dgrad) as the in-order sum of 8 per-group chains, which is what the owner of a row adds up.
On the CPU (gloo tests) the same decomposition runs in torch with fixed (micro-batch, group) GEMM shapes, so 1, 2, 4 and 8 processes issue GEMMs
of the same shapes on the same values.

Per rank and step at P = 8, D = 2048, K = 100352, 24 rows per rank: receives 7 x 9.6 MB of x rows, 2 x 7 x 0.2 MB of y / dy, 7 x 9.6 MB of dx
pieces (~135 MB, ~0.15 ms of xGMI time); computes 1/8 of the head's three GEMMs and updates 1/8 of the weight.
"""
import torch
import torch.distributed as dist

from . import dp

GROUPS = 8


def groups_of(n_out):
    """Canonical groups of the input-gradient sum (isx_head_groups): 8 when the width allows k-tiles of 32 per group, else 1."""
    return GROUPS if n_out % (GROUPS * 32) == 0 else 1


def shardable(weight, world):
    return weight.dim() == 2 and groups_of(weight.size(0)) == GROUPS and world >= 1 and GROUPS % world == 0


def _all_gather_rows(t, group, world):
    if world == 1:
        return t
    out = t.new_empty((world * t.size(0),) + tuple(t.shape[1:]))
    dist.all_gather_into_tensor(out, t.contiguous(), group=group)
    dp.STATS["head_shard_bytes_received"] = dp.STATS.get("head_shard_bytes_received", 0) + t.numel() * 4 * (world - 1)
    return out


class HeadShard(object):
    """One deferred Linear sharded over `world` ranks for the optimizer steps of a training run.  Holds no tensor of its own: the weight stays
    the module's parameter (this rank's rows are the live ones between syncs), the momentum buffer the optimizer's state entry."""

    def __init__(self, weight, bias, group=None):
        self.weight, self.bias, self.group = weight, bias, group
        self.world, self.rank = dp._world(group), dp._rank(group)
        N = weight.size(0)
        self.G = groups_of(N)
        self.Ng = N // self.G
        gpr = self.G // self.world                       # groups per rank
        self.g_lo, self.g_hi = self.rank * gpr, (self.rank + 1) * gpr
        self.lo, self.hi = self.g_lo * self.Ng, self.g_hi * self.Ng
        self.dirty = False                               # rows of other ranks are stale until sync()
        self.calls = []                                  # per head pass of the current step: (leaf ids per rank chunk, X_all, dY_all[:, own])
        self.leaves_per_rank = 0

    def begin_step(self, leaves_per_rank):
        """Every rank runs `leaves_per_rank` micro-batches this step (rank q the consecutive block starting at q * leaves_per_rank)."""
        self.leaves_per_rank, self.calls = int(leaves_per_rank), []

    # ---- primitives -------------------------------------------------------------------------------------------------------------
    def _linear_own(self, X, chunk_rows):
        """X (rows, K) -> (rows, own features).  GPU: the split-K kernel (row-count independent).  CPU: one GEMM per (chunk of chunk_rows rows,
        group): fixed shapes at any world size."""
        W, b = self.weight.detach(), (self.bias.detach() if self.bias is not None else None)
        if X.is_cuda:
            from . import ops
            return ops.head_linear(X, W[self.lo:self.hi], b[self.lo:self.hi] if b is not None else None)
        cols = []
        for g in range(self.g_lo, self.g_hi):
            Wg, bg = W[g * self.Ng:(g + 1) * self.Ng], (b[g * self.Ng:(g + 1) * self.Ng] if b is not None else None)
            cols.append(torch.cat([torch.nn.functional.linear(X[r:r + chunk_rows], Wg, bg) for r in range(0, X.size(0), chunk_rows)], 0))
        return torch.cat(cols, 1)

    def _dgrad_parts(self, dY_own, chunk_rows):
        """dY_own (rows, own features) -> (own groups, rows, K): the per-group chains of the input gradient."""
        W = self.weight.detach()[self.lo:self.hi]
        rows, K = dY_own.size(0), W.size(1)
        if dY_own.is_cuda:
            from ._lib import check, lib
            Mp = (rows + 63) // 64 * 64
            dyT = dY_own.new_zeros((dY_own.size(1), Mp))
            dyT[:, :rows] = dY_own.t()
            parts = torch.empty((self.g_hi - self.g_lo, Mp, K), dtype=torch.float32, device=dY_own.device)
            check(lib().isx_head_linear_dgrad_parts(dyT.data_ptr(), Mp, self.Ng, self.g_hi - self.g_lo, W.data_ptr(), K, parts.data_ptr(),
                                                    torch.cuda.current_stream().cuda_stream), "isx_head_linear_dgrad_parts")
            return parts[:, :rows]
        out = []
        for j in range(self.g_hi - self.g_lo):
            Wg, dg = W[j * self.Ng:(j + 1) * self.Ng], dY_own[:, j * self.Ng:(j + 1) * self.Ng]
            out.append(torch.cat([dg[r:r + chunk_rows].mm(Wg) for r in range(0, rows, chunk_rows)], 0))
        return torch.stack(out, 0)

    # ---- one head pass (forward + backward) ---------------------------------------------------------------------------------------
    def forward(self, x_local, leaf_ids):
        """x_local: this rank's head inputs (R, K) (R equal on every rank); leaf_ids: the global micro-batch indices its rows belong to, in
        order (every rank passes its own; they are exchanged with the rows' counts implied: equal).  Returns (y_local (R, D), ctx)."""
        R = x_local.size(0)
        X = _all_gather_rows(x_local.detach(), self.group, self.world)
        y_own = self._linear_own(X, R)
        if self.world == 1:
            Y = y_own
        else:
            pieces = [torch.empty_like(y_own) for _ in range(self.world)]
            dist.all_gather(pieces, y_own.contiguous(), group=self.group)
            dp.STATS["head_shard_bytes_received"] = dp.STATS.get("head_shard_bytes_received", 0) + y_own.numel() * 4 * (self.world - 1)
            Y = torch.cat(pieces, 1)
        # the micro-batches behind every rank's rows: rank q runs the block that starts leaves_per_rank * (q - rank) after this rank's
        ids = [[i + self.leaves_per_rank * (q - self.rank) for i in leaf_ids] for q in range(self.world)]
        return Y[self.rank * R:(self.rank + 1) * R], {"X": X, "R": R, "ids": ids}

    def backward(self, ctx, dy_local):
        """dy_local (R, D): gradient wrt this rank's head outputs.  Records the rows of this pass for the weight update and returns the gradient
        wrt this rank's head inputs (R, K)."""
        R, X = ctx["R"], ctx["X"]
        dY = _all_gather_rows(dy_local.detach().contiguous(), self.group, self.world)
        self.calls.append((ctx["ids"], X, dY[:, self.lo:self.hi].contiguous()))
        parts = self._dgrad_parts(dY[:, self.lo:self.hi].contiguous(), R)              # (own groups, world * R, K)
        if self.world == 1:
            pieces = parts
        else:
            send = [parts[:, q * R:(q + 1) * R].contiguous() for q in range(self.world)]
            recv = [torch.empty_like(send[0]) for _ in range(self.world)]
            if dist.get_backend(self.group) == "nccl":
                dist.all_to_all(recv, send, group=self.group)
            else:
                _all_to_all_gloo(recv, send, self.group, self.world, self.rank)
            dp.STATS["head_shard_bytes_received"] = dp.STATS.get("head_shard_bytes_received", 0) + send[0].numel() * 4 * (self.world - 1)
            pieces = torch.cat(recv, 0)                                                # rank order = group order
        dx = pieces[0].clone()
        for g in range(1, pieces.size(0)):
            dx += pieces[g]                                                            # the canonical in-order sum of the group chains
        return dx

    # ---- end of the optimizer step --------------------------------------------------------------------------------------------------
    def rows_in_canonical_order(self):
        """(X, dY_own) of the whole mini-batch, rows in micro-batch order (the order a single process produces them in)."""
        chunks = []
        for ids, X, dY in self.calls:
            R = X.size(0) // len(ids)
            for q, leaf_ids in enumerate(ids):
                per = R // max(len(leaf_ids), 1)
                for j, leaf in enumerate(leaf_ids):
                    a = q * R + j * per
                    chunks.append((leaf, X[a:a + per], dY[a:a + per]))
        chunks.sort(key=lambda c: c[0])
        self.calls = []
        if not chunks:
            K = self.weight.size(1)
            return self.weight.new_zeros((0, K)), self.weight.new_zeros((0, self.hi - self.lo))
        return torch.cat([c[1] for c in chunks], 0), torch.cat([c[2] for c in chunks], 0)

    def finish(self, optimizer):
        """dW of this rank's rows over the rows of the whole mini-batch + the SGD update of those rows (one kernel on the GPU)."""
        X, dY = self.rows_in_canonical_order()
        sgd_update_rows(optimizer, self.weight, dY, X, self.lo, self.hi, self.Ng)
        self.dirty = self.world > 1

    def sync(self):
        """Every rank's rows of the weight to every rank (before anything reads the whole weight: embedding pass, evaluation, checkpoint)."""
        if not self.dirty:
            return
        with torch.no_grad():
            w = self.weight.data
            mine = w[self.lo:self.hi].clone()
            dist.all_gather_into_tensor(w.view(-1), mine.view(-1), group=self.group)
            dp.STATS["head_shard_sync_bytes_received"] = mine.numel() * 4 * (self.world - 1)
        dp.bump_version(self.weight)
        self.dirty = False


def _all_to_all_gloo(recv, send, group, world, rank):
    """gloo has no all_to_all: `world` rounds of all_gather on the piece every rank holds for rank q."""
    for q in range(world):
        got = [torch.empty_like(send[q]) for _ in range(world)]
        dist.all_gather(got, send[q], group=group)
        if q == rank:
            for s in range(world):
                recv[s].copy_(got[s])


def sgd_update_rows(optimizer, w, dY, X, lo, hi, group_rows):
    """Rows [lo, hi) of `w`: gradient dY^T X over the given rows + torch.optim.SGD's update, the momentum kept in the optimizer's state entry of
    `w` (a full-size buffer of which this rank maintains its rows).  GPU: isx_head_sgd_step on the row slice.  CPU: the gradient as one GEMM per
    group of `group_rows` output features (fixed shapes at any world size), the update in separate torch ops (no fused multiply-add)."""
    if type(optimizer) is not torch.optim.SGD:
        raise TypeError("a sharded head is updated by its own SGD kernel: the training optimizer must be torch.optim.SGD, got %s" % type(optimizer).__name__)
    group = next((g for g in optimizer.param_groups if any(p is w for p in g['params'])), None)
    if group is None or group.get('maximize'):
        raise ValueError("the sharded head's weight is not a (minimised) parameter of the optimizer")
    mom, wd, lr, damp, nest = float(group['momentum']), float(group['weight_decay']), float(group['lr']), float(group['dampening']), bool(group['nesterov'])
    first, buf = False, None
    if mom != 0.0:
        state = optimizer.state[w]
        buf = state.get('momentum_buffer')
        if buf is None:
            buf = state['momentum_buffer'] = torch.zeros_like(w)
            first = True
    with torch.no_grad():
        ws = w.data[lo:hi]
        bs = buf[lo:hi] if buf is not None else None
        if w.is_cuda and (hi - lo) % 64 == 0 and w.size(1) % 128 == 0 and 128 * w.size(1) * 4 < 2 ** 31:
            from ._lib import check, lib
            check(lib().isx_head_sgd_step(dY.contiguous().data_ptr(), X.contiguous().data_ptr(), X.size(0), hi - lo, w.size(1), ws.data_ptr(),
                                          bs.data_ptr() if bs is not None else None, 1 if first else 0, lr, mom, damp, wd, 1 if nest else 0,
                                          torch.cuda.current_stream().cuda_stream), "isx_head_sgd_step")
        else:
            for a in range(0, hi - lo, group_rows):
                g = dY[:, a:a + group_rows].t().mm(X) if X.size(0) else ws.new_zeros((group_rows, w.size(1)))
                wg = ws[a:a + group_rows]
                if wd != 0.0:
                    g = g + wd * wg
                upd = g
                if mom != 0.0:
                    bg = bs[a:a + group_rows]
                    bg.copy_(g if first else mom * bg + (1.0 - damp) * g)
                    upd = g + mom * bg if nest else bg
                wg.sub_(lr * upd)
    w.grad = None
    dp.bump_version(w)
