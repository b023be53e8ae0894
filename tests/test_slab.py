"""Descriptor-slab file format (isx/slab.py): round trip, row ranges (= gallery shards), errors."""
import os

import numpy as np
import pytest
import torch

from isx import slab
from isx.retrieval import shard_bounds


def test_round_trip_and_shards(tmp_path):
    g = torch.Generator().manual_seed(0)
    E = torch.nn.functional.normalize(torch.randn(1003, 48, generator=g), dim=1)
    lab = torch.arange(1003, dtype=torch.int32) % 17
    p = str(tmp_path / "gallery.isxslab")
    slab.save_slab(p, E, lab)
    info = slab.slab_info(p)
    assert info == {"rows": 1003, "dim": 48, "normalised": True, "has_labels": True, "label_offset": info["label_offset"]}
    assert info["label_offset"] % 4096 == 0
    e2, l2 = slab.load_slab(p)
    assert torch.equal(e2, E) and torch.equal(l2, lab)
    parts = [slab.load_slab(p, rows=shard_bounds(1003, 8, r)) for r in range(8)]
    assert torch.equal(torch.cat([x[0] for x in parts]), E) and torch.equal(torch.cat([x[1] for x in parts]), lab)
    e0, l0 = slab.load_slab(p, rows=(5, 5))
    assert e0.shape == (0, 48) and l0.shape == (0,)
    slab.save_slab(p, E.numpy(), None, normalised=False)
    e3, l3 = slab.load_slab(p)
    assert l3 is None and torch.equal(e3, E) and not slab.slab_info(p)["normalised"]


def test_errors(tmp_path):
    p = str(tmp_path / "x.bin")
    open(p, "wb").write(b"\0" * 8192)
    with pytest.raises(ValueError):
        slab.slab_info(p)
    q = str(tmp_path / "y.isxslab")
    slab.save_slab(q, np.zeros((4, 8), np.float32))
    with pytest.raises(ValueError):
        slab.load_slab(q, rows=(2, 9))
    with pytest.raises(ValueError):
        slab.save_slab(q, np.zeros((4, 8), np.float32), np.zeros((3,), np.int32))


@pytest.mark.gpu
def test_gpu_load_streams_through_pinned_staging(tmp_path):
    E = torch.randn(70000, 64)
    p = str(tmp_path / "g.isxslab")
    slab.save_slab(p, E, torch.arange(70000, dtype=torch.int32))
    e, l = slab.load_slab(p, device="cuda", rows=(100, 69000))
    assert e.is_cuda and torch.equal(e.cpu(), E[100:69000]) and torch.equal(l.cpu(), torch.arange(100, 69000, dtype=torch.int32))


def test_slab_writer_streams_row_blocks(tmp_path):
    """SlabWriter: header first, row blocks appended from tensors / arrays, labels at close; a short or long stream is refused and leaves no file."""
    from isx import slab
    d = torch.randn(300, 24)
    p = str(tmp_path / "s.slab")
    w = slab.SlabWriter(p, 300, 24, has_labels=True)
    w.append(d[:100]); w.append(d[100:120].numpy()); w.append(d[120:])
    w.close(torch.arange(300, dtype=torch.int32))
    got, lab = slab.load_slab(p)
    assert torch.equal(got, d) and torch.equal(lab, torch.arange(300, dtype=torch.int32))
    w = slab.SlabWriter(str(tmp_path / "t.slab"), 10, 24)
    w.append(d[:4])
    with pytest.raises(ValueError):
        w.close()
    assert not os.path.exists(str(tmp_path / "t.slab")) and not os.path.exists(str(tmp_path / "t.slab.tmp"))
    ref = [(None, "l%d" % (i % 3), "p%d" % i) for i in range(300)]
    slab.save_gallery(p, d, ref, ["l0", "l1", "l2"])
    g, rs, names = slab.load_gallery(p, rows=(10, 20))
    assert torch.equal(g, d[10:20]) and rs[0] == (None, "l1", "p10") and names == ["l0", "l1", "l2"]
