"""The decoder processes in front of the extraction path (train/_decode_farm.py): what they return is what the in-process read
returns (reference utils/image.py:211-214 -- the file as 8-bit RGB), for every way the ingest calls them."""
import glob
import os
import signal
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))


@pytest.fixture()
def farm_env(monkeypatch):
    from train import _decode_farm as DF
    monkeypatch.setenv("ISX_DECODE_PROCS", "3")
    DF.shutdown()
    yield DF
    DF.shutdown()
    assert not glob.glob("/dev/shm/isx_decode_%d_*" % os.getpid())          # nothing left behind


def _write(tmp_path, n, size=(40, 56), fmt="jpg"):
    from PIL import Image
    files = []
    for i in range(n):
        a = np.random.default_rng(i).integers(0, 256, size + (3,), dtype=np.uint8)
        f = str(tmp_path / ("im%04d.%s" % (i, fmt)))
        Image.fromarray(a).save(f, quality=90) if fmt == "jpg" else Image.fromarray(a).save(f)
        files.append(f)
    return files


def test_farm_equals_in_process_read(tmp_path, farm_env):
    from test import _common as C
    files = _write(tmp_path, 70) + _write(tmp_path / "..", 5, fmt="png")
    from PIL import Image
    grey = str(tmp_path / "grey.png")
    Image.fromarray(np.arange(40 * 56, dtype=np.uint8).reshape(40, 56)).save(grey)      # a single-channel file is read as RGB, as imread_rgb does
    files.append(grey)
    load = C.ImageLoader(raw=True)
    got = farm_env.decode_files(files, 40 * 56 * 3, window=16)
    assert len(got) == len(files)
    for f, t in zip(files, got):
        assert t.dtype == torch.uint8 and torch.equal(t, load(f)), f


def test_slots_are_recycled_and_segments_grow(tmp_path, farm_env, monkeypatch):
    monkeypatch.setattr(farm_env, "_SEGMENT_BYTES", 16 * 40 * 56 * 3)                   # 16 slots per segment
    files = _write(tmp_path, 8)
    farm = farm_env.decode_farm()
    want = [torch.from_numpy(np.asarray(__import__("PIL.Image").Image.open(f).convert("RGB")).copy()) for f in files]
    held = farm.submit(files * 5, 40 * 56 * 3)                                          # 40 tickets held at once: three segments
    assert len(farm.by_id) == 3
    for k, t in enumerate(held):
        assert torch.equal(t.tensor(), want[k % 8])
    for t in held:
        t.release()
    again = farm.submit(files * 5, 40 * 56 * 3)                                          # served from the freed slots
    assert len(farm.by_id) == 3 and all(torch.equal(t.tensor(), want[k % 8]) for k, t in enumerate(again))
    for t in again:
        t.release()
    assert all(seg.path is None for seg in farm.by_id.values())                         # every worker mapped them: the names are gone


def test_errors_surface_at_the_image(tmp_path, farm_env):
    files = _write(tmp_path, 4)
    bad = str(tmp_path / "bad.jpg")
    open(bad, "wb").write(b"not an image")
    tickets = farm_env.decode_farm().submit(files[:2] + [bad] + files[2:], 40 * 56 * 3)
    with pytest.raises(farm_env.DecodeError, match="bad.jpg"):
        tickets[2].tensor()
    assert tickets[3].tensor().shape == (40, 56, 3)                                     # its neighbours are fine
    for t in tickets:
        t.release()
    with pytest.raises(farm_env.DecodeError, match="line breaks"):
        farm_env.decode_farm().submit(["a\nb.jpg"], 64)


def test_a_file_larger_than_the_slot_is_still_read(tmp_path, farm_env):
    big = _write(tmp_path, 1, size=(90, 120))[0]
    small = _write(tmp_path / "..", 1)[0]
    from test import _common as C
    out = farm_env.decode_files([big, small], 40 * 56 * 3)
    assert out[0].shape == (90, 120, 3) and torch.equal(out[0], C.ImageLoader(raw=True)(big)) and out[1].shape == (40, 56, 3)


def test_a_dead_decoder_fails_loudly(tmp_path, farm_env):
    """One decoder killed: what it held fails at the image, the others keep serving; all killed: a DecodeError, not a hang."""
    import time
    files = _write(tmp_path, 6)
    farm = farm_env.decode_farm()
    assert all(t.tensor().shape == (40, 56, 3) for t in farm.submit(files, 40 * 56 * 3))
    os.kill(farm.procs[0].pid, signal.SIGKILL)
    farm.procs[0].wait()
    for _ in range(100):
        if farm.dead[0]:
            break
        time.sleep(0.05)
    assert farm.dead[0]
    tickets = farm.submit(files * 4, 40 * 56 * 3)                 # the two survivors take everything
    assert all(t.chunk.worker != 0 and t.tensor().shape == (40, 56, 3) for t in tickets)
    for t in tickets:
        t.release()
    for p in farm.procs[1:]:
        os.kill(p.pid, signal.SIGKILL)
        p.wait()
    with pytest.raises(farm_env.DecodeError):
        for t in farm.submit(files, 40 * 56 * 3):
            t.tensor()


def test_lazy_images_through_the_farm(tmp_path, farm_env):
    """LazyImage entries with the plain RGB loader: prefetch_all hands the batch to the processes, get() is a view checked against the set's
    shape, resolve_images copies out before the slots are recycled; ISX_DECODE_PROCS=0 serves the same entries from the thread pool."""
    from test import _common as C
    from train import _common as TC
    files = _write(tmp_path, 12)
    load = C.ImageLoader(raw=True)
    want = [load(f) for f in files]
    lazies = [TC.LazyImage(f, load, (40, 56, 3), torch.uint8) for f in files]
    TC.prefetch_all(lazies)
    assert all(im.shared() for im in lazies)
    got = TC.resolve_images(lazies)
    assert all(torch.equal(a, b) for a, b in zip(got, want)) and all(im._fut is None for im in lazies)
    more = TC.resolve_images([TC.LazyImage(f, load, (40, 56, 3), torch.uint8) for f in reversed(files)])      # the slots just freed are written again
    assert all(torch.equal(a, b) for a, b in zip(got, want)) and all(torch.equal(a, b) for a, b in zip(more, reversed(want)))
    odd = TC.LazyImage(_write(tmp_path / "..", 1, size=(30, 30))[0], load, (40, 56, 3), torch.uint8)
    with pytest.raises(RuntimeError, match="same-sized"):
        odd.get()
    odd.release()
    os.environ["ISX_DECODE_PROCS"] = "0"
    lz = TC.LazyImage(files[0], load, (40, 56, 3), torch.uint8)
    lz.prefetch()
    assert not lz.shared() and torch.equal(lz.get(), want[0])


def test_decode_all_uses_the_farm_for_both_loader_forms(tmp_path, farm_env, monkeypatch):
    from test import _common as C
    monkeypatch.setattr(C, "FARM_MIN_FILES", 8)
    monkeypatch.setenv("ISX_DECODE_THREADS", "4")
    files = _write(tmp_path, 20)
    raw, norm = C.ImageLoader(raw=True), C.ImageLoader([0.4, 0.5, 0.6], [0.2, 0.3, 0.25])
    got_raw, got_norm = C._decode_all(raw, files), C._decode_all(norm, files)
    assert farm_env._FARM is not None and farm_env._FARM.next_id > 0                     # the files did go through the processes
    for f, a, b in zip(files, got_raw, got_norm):
        assert torch.equal(a, raw(f)) and torch.equal(b, norm(f)) and b.shape == (3, 40, 56) and b.dtype == torch.float32
