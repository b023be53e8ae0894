"""Batch fold driver (reference utils/train_general.py:27-38).  The training loop helpers of
that file (train_gen, anneal, micro/mini batches) are out of scope."""


def fold_batches(f, init, x, batch_size, cut_end=False, add_args={}):
    """Left fold of `f(acc, start_index, is_final, x[start:end], **add_args)` over consecutive
    batches of `x`.  batch_size <= 0: one call on the whole set.  cut_end drops a trailing
    partial batch (and then flags the last FULL batch as final)."""
    n = len(x)
    if batch_size <= 0:
        return f(init, 0, True, x, **add_args)
    acc = init
    for start in range(0, n, batch_size):
        end = min(start + batch_size, n)
        if cut_end and start + batch_size > n:
            continue
        is_final = (end > n - batch_size) if cut_end else (end == n)
        acc = f(acc, start, is_final, x[start:end], **add_args)
    return acc
