"""Retrieval metrics with the reference's signatures (utils/metrics.py:8-55):
precision1, avg_precision, mean_avg_precision over a (queries x gallery) similarity matrix
and two datasets of (tensor, label, path) tuples.

The reference sorts every row and walks every rank in a Python loop (16 ms per query at
N = 10k).  Here a GPU matrix goes through libisx: `isx_topk_rows` for P@k,
`isx_rank_full` + `isx_average_precision` for AP -- canonical order (score desc, index asc),
float64 AP accumulated rank by rank in the same operation order as the Python loop, so on
the same score matrix the values are bit-identical to the reference's.  A CPU matrix (the
reference's `--device=-1` path) is handled with the same arithmetic in vectorised torch.
"""
import torch


def _label_ids(test_set, ref_set):
    table = {}
    ids = lambda ds: [table.setdefault(lab, len(table)) for _, lab, _ in ds]
    q, g = ids(test_set), ids(ref_set)
    return torch.tensor(q, dtype=torch.int32), torch.tensor(g, dtype=torch.int32)


def _topk(sim, k):
    """Canonical top-k (values, indices) of every row."""
    if sim.is_cuda:
        from isx import ops
        return ops.topk_rows(sim.float(), k)
    order = sim.sort(dim=1, descending=True, stable=True)
    return order.values[:, :k], order.indices[:, :k]


def precision1(sim, test_set, ref_set, kth=1):
    """Label of the kth-ranked gallery item vs the query label.
    Returns (precision, correct, total, max_sim (M,1), max_label list)."""
    total = sim.size(0)
    kth = max(kth, 1)
    vals, idx = _topk(sim, kth)
    hit_idx = idx[:, kth - 1].tolist()
    max_label = [ref_set[j][1] for j in hit_idx]
    correct = sum(1 for (_, lab, _), got in zip(test_set, max_label) if lab == got)
    return float(correct) / total, correct, total, vals[:, kth - 1:kth], max_label


def _average_precisions(sim, qlab, glab, kth):
    """float64 AP per query, NaN where the query has no (remaining) positive."""
    if sim.is_cuda:
        from isx import ops
        # no full sort: AP needs only the ranks of the positives (isx_average_precision_sim); queries
        # with many positives fall back to isx_rank_full + isx_average_precision inside the wrapper
        return ops.average_precision_sim(sim.float(), qlab.to(sim.device), glab.to(sim.device), kth).cpu()
    M, N = sim.shape
    ranked = sim.sort(dim=1, descending=True, stable=True).indices
    n_pos = (glab[None, :] == qlab[:, None]).sum(1) - (kth - 1)
    hit = (glab[ranked.long()] == qlab[:, None])[:, kth - 1:].double()          # the first kth-1 ranks are ignored
    j = torch.arange(hit.size(1), dtype=torch.float64)[None, :]
    incl = hit.cumsum(1)
    before = incl - hit
    dn = n_pos.clamp(min=1).double()[:, None]
    recall, old_recall = incl / dn, before / dn
    precision = incl / (j + 1.0)
    old_precision = torch.where(j == 0, torch.ones_like(j), before / j.clamp(min=1.0))
    terms = (recall - old_recall) * ((old_precision + precision) / 2.0)            # exactly 0 off the hits
    ap = terms.cumsum(1)[:, -1] if hit.size(1) > 0 else torch.zeros(M, dtype=torch.float64)   # sequential, rank order
    return torch.where(n_pos > 0, ap, torch.full_like(ap, float('nan')))


def avg_precision(sim, i, test_set, ref_set, kth=1):
    """Oxford-buildings AP of query i; None when it has no positive left after skipping kth-1."""
    qlab, glab = _label_ids(test_set, ref_set)
    ap = _average_precisions(sim[i:i + 1], qlab[i:i + 1], glab, kth)[0].item()
    return None if ap != ap else ap


def mean_avg_precision(sim, test_set, ref_set, kth=1):
    qlab, glab = _label_ids(test_set, ref_set)
    aps = [a for a in _average_precisions(sim, qlab, glab, kth).tolist() if a == a]
    return sum(aps) / float(len(aps))


# ---- evaluation without the whole score matrix -----------------------------------------------------------------
# P@k, AP and the masked score sums are all PER QUERY ROW: a (queries x gallery) matrix larger than the budget is
# evaluated in blocks of query rows, each block's scores materialised, consumed and dropped.  Same kernels on the same
# rows -> the same bits as the one-matrix evaluation, at any gallery size (10 k x 1 M fp32 would be 40 GB; in blocks of
# 256 queries it is 1 GB at a time).  The reference has no counterpart: it builds the full matrix (utils/metrics.py:25-55
# consume it) and falls back to the CPU when it does not fit (utils/train_siamese.py:30-43).
SIM_BUDGET_BYTES = 8 << 30


def row_blocks(M, N, budget_bytes=None):
    """[(r0, r1)] covering range(M) with (r1 - r0) * N * 4 <= budget (at least one row per block)."""
    budget = SIM_BUDGET_BYTES if budget_bytes is None else budget_bytes
    per = max(1, int(budget // max(1, 4 * N)))
    if per >= M:
        return [(0, M)] if M else []
    if per >= 128:
        per = per // 128 * 128                 # whole GEMM tiles
    return [(r, min(r + per, M)) for r in range(0, M, per)]


def retrieval_metrics(test_emb, ref_emb, test_set, ref_set, kth=1, budget_bytes=None, with_sums=False):
    """precision1(...) + mean_avg_precision(...) (+ the pos / all score sums of test_descriptor_net) of
    sim = test_emb @ ref_emb.T, evaluated in query-row blocks.  Returns a dict: prec1, correct, total, max_sim (M,1),
    max_label, mAP, sum_pos, sum_all, blocks."""
    from .train_siamese import similarity_matrix
    M, N = test_emb.size(0), ref_emb.size(0)
    qlab, glab = _label_ids(test_set, ref_set)
    blocks = row_blocks(M, N, budget_bytes)
    correct, max_sims, max_label, aps = 0, [], [], []
    sum_pos, sum_all = 0.0, 0.0
    for r0, r1 in blocks:
        sim = similarity_matrix(test_emb[r0:r1], ref_emb)
        _, c, _, ms, ml = precision1(sim, test_set[r0:r1], ref_set, kth)
        correct += c
        max_sims.append(ms)
        max_label.extend(ml)
        aps.extend(_average_precisions(sim, qlab[r0:r1], glab, kth).tolist())
        if with_sums:
            if sim.is_cuda:
                from isx import ops
                rows = ops.masked_sums(sim, qlab[r0:r1].cuda(), glab.cuda()).cpu()
                for a, b in rows.tolist():              # row order, as the one-matrix evaluation adds them
                    sum_pos += a
                    sum_all += b
            else:
                mask = qlab[r0:r1][:, None] == glab[None, :]
                sum_pos += float(sim[mask].double().sum())
                sum_all += float(sim.double().sum())
        del sim
    valid = [a for a in aps if a == a]
    return {"prec1": float(correct) / max(M, 1), "correct": correct, "total": M, "max_sim": torch.cat(max_sims, 0) if max_sims else None, "max_label": max_label,
            "mAP": sum(valid) / float(len(valid)) if valid else float("nan"), "aps": aps, "sum_pos": sum_pos, "sum_all": sum_all, "blocks": len(blocks)}


# ---- average precision against a gallery sharded by rows -------------------------------------------------------------------------
# The reference ranks ONE full score row per query (utils/metrics.py:25-45).  AP only depends on the ranks of the positives, and a rank is a count
# of gallery keys, which adds over shards: three steps around two collectives (include/isx.h, isx_ap_shard_*).  GPU tensors run libisx's kernels;
# CPU tensors (the reference's --device=-1 mode, the gloo tests) the numpy restatement below -- the same integers, the same float64 expressions.
_AP_MAXP = 2048


def _cpu_rank_keys(scores, gidx):
    """Canonical ranking keys (larger = ranked earlier: score descending, global index ascending; -0.0 folded onto +0.0) as uint64."""
    import numpy as np
    s = np.array(scores, dtype=np.float32, copy=True)
    s[s == 0.0] = 0.0
    u = s.view(np.uint32)
    ob = np.where(u & np.uint32(0x80000000), ~u, u | np.uint32(0x80000000)).astype(np.uint64)
    return (ob << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - np.asarray(gidx, dtype=np.uint64))


def ap_shard_positives(sim, idx_base, qlab, glab, cap=_AP_MAXP):
    """(keys (M, cap) int64 bit patterns of the uint64 keys, 0 = empty; count (M,) int32): this shard's positives per query."""
    if sim.is_cuda:
        from isx import ops
        return ops.ap_shard_positives(sim.float(), idx_base, qlab.to(sim.device), glab.to(sim.device), cap)
    import numpy as np
    s, q, g = sim.float().numpy(), qlab.numpy(), glab.numpy()
    M = s.shape[0]
    keys, count = np.zeros((M, cap), dtype=np.uint64), np.zeros((M,), dtype=np.int32)
    for i in range(M):
        j = np.nonzero(g == q[i])[0]
        count[i] = len(j)
        k = _cpu_rank_keys(s[i, j], idx_base + j)[:cap]
        keys[i, :len(k)] = k
    return torch.from_numpy(keys.view(np.int64)), torch.from_numpy(count)


def ap_shard_hist(sim, idx_base, keys_all):
    """This shard's bucket counts (M, 2048) int32 against the gathered keys of all shards (M, W)."""
    if sim.is_cuda:
        from isx import ops
        return ops.ap_shard_hist(sim.float(), idx_base, keys_all.to(sim.device))
    import numpy as np
    s, ka = sim.float().numpy(), keys_all.numpy().view(np.uint64)
    M, N = s.shape
    hist = np.zeros((M, _AP_MAXP), dtype=np.int32)
    gidx = idx_base + np.arange(N)
    for i in range(M):
        pk = np.sort(ka[i][ka[i] != 0])                          # ascending
        if len(pk) == 0 or len(pk) > _AP_MAXP:
            continue
        x = _cpu_rank_keys(s[i], gidx)
        x = x[x >= pk[0]]
        b = len(pk) - np.searchsorted(pk, x, side='right')       # #{positives > x}
        hist[i, :len(pk)] = np.bincount(b, minlength=len(pk))[:len(pk)]
    return torch.from_numpy(hist)


def ap_from_hist(hist, n_lab, kth=1):
    """float64 AP per query from the histogram summed over the shards; NaN without a (remaining) positive, -1 over 2048 positives."""
    if hist.is_cuda:
        from isx import ops
        return ops.ap_from_hist(hist, n_lab.to(hist.device), kth)
    import numpy as np
    h, nl = hist.numpy(), n_lab.numpy()
    out = np.empty((h.shape[0],), dtype=np.float64)
    for r in range(h.shape[0]):
        n_lab_r = int(nl[r])
        n_pos = n_lab_r - (kth - 1)
        if n_pos <= 0:
            out[r] = float('nan')
            continue
        if n_lab_r > _AP_MAXP or n_lab_r > h.shape[1]:
            out[r] = -1.0
            continue
        dn, ap, before, hh = float(n_pos), 0.0, 0, 0
        for i in range(n_lab_r):
            before += int(h[r, i])
            rp = before - 1
            if rp < kth - 1:
                continue
            hcur, hh = hh, hh + 1
            j = rp - (kth - 1)
            recall, old_recall = float(hcur + 1) / dn, float(hcur) / dn
            precision = float(hcur + 1) / (float(j) + 1.0)
            old_precision = 1.0 if j == 0 else float(hcur) / float(j)
            ap += (recall - old_recall) * ((old_precision + precision) / 2.0)
        out[r] = ap
    return torch.from_numpy(out)


def sharded_average_precisions(test_emb, shard_emb, idx_base, qlab, glab_shard, kth=1, group=None, budget_bytes=None):
    """AP (float64, (M,)) of every query against a gallery whose rows are sharded over the ranks of `group`: `shard_emb` are this rank's rows
    (global rows idx_base ...), `glab_shard` their labels, `test_emb` / `qlab` the replicated queries.  The same values on every rank, and the
    same float64 bits as the unsharded evaluation.  Query rows go block by block (the (block, shard) score matrix within budget_bytes)."""
    import torch.distributed as dist
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    M, Ns = test_emb.size(0), shard_emb.size(0)
    dev = test_emb.device
    on_wire = (lambda t: t) if world == 1 or dist.get_backend(group) == 'nccl' else (lambda t: t.cpu())
    out = []
    ns_max = Ns
    if world > 1:                                                # the query blocks must be the same on every rank: sized by the LARGEST shard
        t = on_wire(torch.tensor([Ns], dtype=torch.int64, device=dev))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
        ns_max = int(t)
    for r0, r1 in row_blocks(M, max(ns_max, 1), budget_bytes):
        from .train_siamese import similarity_matrix
        sim = similarity_matrix(test_emb[r0:r1], shard_emb) if Ns else test_emb.new_zeros((r1 - r0, 0))
        keys, count = ap_shard_positives(sim, idx_base, qlab[r0:r1], glab_shard, _AP_MAXP)
        if world > 1:
            cap = on_wire(count.max().reshape(1).to(torch.int64)) if count.numel() else torch.zeros(1, dtype=torch.int64)
            dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
            cap = max(1, min(int(cap), _AP_MAXP))
            mine = on_wire(keys[:, :cap].contiguous())
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=group)
            keys_all = torch.cat(parts, 1).to(dev)
            n_lab = on_wire(count.clone())
            dist.all_reduce(n_lab, group=group)
            n_lab = n_lab.to(dev)
        else:
            keys_all, n_lab = keys, count
        hist = ap_shard_hist(sim, idx_base, keys_all)
        ld = max(1, min(int(n_lab.max()) if n_lab.numel() else 1, _AP_MAXP))
        hist = hist[:, :ld].contiguous()                         # the buckets past the largest number of positives are zero: they need not travel
        if world > 1:
            h = on_wire(hist)
            dist.all_reduce(h, group=group)
            hist = h.to(dev)
        out.append(ap_from_hist(hist, n_lab, kth).cpu())
        del sim
    return torch.cat(out, 0) if out else torch.zeros((0,), dtype=torch.float64)
