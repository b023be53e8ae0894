"""Image files -> (H, W, 3) uint8 tensors on a farm of decoder PROCESSES (the step in front of the extraction path).

Why processes: PIL's JPEG decoder drops the GIL only inside libjpeg; opening the file, feeding it in chunks, converting and wrapping the result are
Python, and at 224 x 224 (0.56 ms of decode per image) a pool of threads spends its time handing the GIL around -- 8 threads decode FEWER images
per second than one (1.0 k against 1.8 k on this container's 8 cores); 8 processes decode 14.8 k.  The reference decodes one file at a time on
the main thread (test/classif_finetune_test.py:62-73).

Layout: one farm per process (decode_farm()), `workers` children running train/_decode_worker.py (stand-alone, no torch, no GPU), each on a pipe
pair.  Pixels come back through shared files under /dev/shm mapped by both sides: a SEGMENT is a file of equal slots (slot bytes = the size of
the images being read); the parent hands a worker a chunk of (slot offset, file name) pairs, the worker decodes each file into its slot and answers
one line.  A segment's file is unlinked as soon as every worker has mapped it, so a killed run leaves nothing behind in /dev/shm.  Segments are
added when the free slots run out (the consumer of a lazy set holds at most the batches it decodes ahead), never while tickets wait: no deadlock.

A Ticket stands for one file: tensor() blocks until its chunk is back and returns a view of the slot (no copy), release() gives the slot back.
Errors of the decoder are raised by tensor(), at the image they belong to.  A worker that dies fails every ticket it held, loudly."""
import atexit
import mmap
import os
import subprocess
import sys
import tempfile
import threading

import numpy as np
import torch

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_decode_worker.py")
_SEGMENT_BYTES = 256 << 20
_CHUNK = 8                     # files per request line


class DecodeError(RuntimeError):
    pass


class _Segment(object):
    def __init__(self, sid, slot_bytes, slots, n_workers):
        self.sid, self.slot_bytes, self.slots = sid, slot_bytes, slots
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
        fd, self.path = tempfile.mkstemp(prefix="isx_decode_%d_" % os.getpid(), dir=d)
        try:
            os.ftruncate(fd, slot_bytes * slots)
            self.mm = mmap.mmap(fd, slot_bytes * slots)
        finally:
            os.close(fd)
        self.arr = np.frombuffer(self.mm, dtype=np.uint8)
        self.free = list(range(slots - 1, -1, -1))
        self.unmapped_by = n_workers            # workers that have not answered the attach yet; 0 -> the file name is dropped

    def unlink(self):
        if self.path is not None:
            try:
                os.unlink(self.path)
            except OSError:
                pass
            self.path = None


class _Chunk(object):
    __slots__ = ("done", "results", "error", "worker", "n")

    def __init__(self, worker, n):
        self.done, self.results, self.error, self.worker, self.n = threading.Event(), None, None, worker, n


class Ticket(object):
    __slots__ = ("farm", "seg", "slot", "chunk", "k", "path", "_released")

    def __init__(self, farm, seg, slot, chunk, k, path):
        self.farm, self.seg, self.slot, self.chunk, self.k, self.path, self._released = farm, seg, slot, chunk, k, path, False

    def tensor(self):
        """(H, W, 3) uint8 view of the slot.  Valid until release()."""
        self.chunk.done.wait()
        if self.chunk.error is not None:
            raise DecodeError("decoder process failed while reading %s: %s" % (self.path, self.chunk.error))
        st, h, w, msg = self.chunk.results[self.k]
        if st == "e":
            raise DecodeError("%s: %s" % (self.path, msg))
        if st == "b":                                     # larger than the slot (a ragged folder read up front): decoded here, same call as the worker's
            from PIL import Image
            return torch.from_numpy(np.asarray(Image.open(self.path).convert("RGB"), dtype=np.uint8).copy())
        off = self.slot * self.seg.slot_bytes
        return torch.from_numpy(self.seg.arr[off:off + h * w * 3].reshape(h, w, 3))

    def result(self):
        return self.tensor()

    def copy(self):
        """The image as a tensor of its own (one memcpy on this thread).  Not tensor().clone(): torch copies 150 KB with its OpenMP team, and a
        team of 16 that has to assemble while 14 decoder processes keep every core busy takes a scheduler quantum per copy -- 1 ms per image
        measured on the MI355X box, 2.3 of the 4.6 s of an evaluation run on 22 000 files."""
        t = self.tensor()
        return torch.from_numpy(t.numpy().copy())

    def release(self):
        if not self._released:
            self._released = True
            self.chunk.done.wait()
            self.farm._give_back(self.seg, self.slot)


class DecodeFarm(object):
    def __init__(self, workers):
        self.n = max(1, int(workers))
        self.lock = threading.Lock()
        self.segments = {}                                # slot bytes -> [segments]
        self.by_id = {}
        self.pending = {}                                 # request id -> _Chunk
        self.load = [0] * self.n                          # files outstanding per worker
        self.dead = [False] * self.n                      # the worker's answer pipe closed: nothing more is handed to it
        self.next_id = 0
        self.closed = False
        self.pid = os.getpid()
        self.submit_lock = threading.Lock()               # a request must not overtake the attach message of the segment it names
        env = dict(os.environ)
        env.pop("PYTHONSTARTUP", None)
        self.procs = [subprocess.Popen([sys.executable, "-u", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env, close_fds=True)
                      for _ in range(self.n)]
        self.wlocks = [threading.Lock() for _ in range(self.n)]
        self.readers = [threading.Thread(target=self._read, args=(i,), daemon=True, name="isx-decode-%d" % i) for i in range(self.n)]
        for t in self.readers:
            t.start()
        atexit.register(self.close)

    # ---- the answers of worker i ----
    def _read(self, i):
        p = self.procs[i]
        try:
            for line in p.stdout:
                kind, _, rest = line.rstrip(b"\n").partition(b"\t")
                if kind == b"A":
                    with self.lock:
                        seg = self.by_id.get(int(rest))
                        if seg is not None:
                            seg.unmapped_by -= 1
                            if seg.unmapped_by == 0:
                                seg.unlink()
                    continue
                rid, _, body = rest.partition(b"\t")
                res = []
                for item in body.split(b";"):
                    f = item.split(b" ", 3)
                    res.append((f[0].decode(), int(f[1]), int(f[2]), f[3].decode("utf-8", "replace") if len(f) > 3 else ""))
                with self.lock:
                    ch = self.pending.pop(int(rid), None)
                    if ch is not None:
                        self.load[i] -= ch.n
                if ch is not None:
                    ch.results = res
                    ch.done.set()
        finally:                                           # the pipe closed: close() or the worker died -- nobody waits for ever
            with self.lock:
                self.dead[i] = True
                dead = [(r, c) for r, c in self.pending.items() if c.worker == i]
                for r, _ in dead:
                    del self.pending[r]
            for _, c in dead:
                c.error = "decoder process %d ended (exit code %s)" % (i, p.poll())
                c.done.set()

    def _send(self, i, data):
        with self.wlocks[i]:
            try:
                self.procs[i].stdin.write(data)
                self.procs[i].stdin.flush()
            except (BrokenPipeError, OSError, ValueError) as e:
                raise DecodeError("decoder process %d is gone: %s" % (i, e))

    # ---- slots ----
    def _new_segment(self, slot_bytes):
        slots = int(max(16, min(1024, _SEGMENT_BYTES // slot_bytes)))
        sid = len(self.by_id)
        seg = _Segment(sid, slot_bytes, slots, self.n)
        self.by_id[sid] = seg
        self.segments.setdefault(slot_bytes, []).append(seg)
        return seg

    def _take(self, slot_bytes, count):
        """`count` (segment, slot) pairs; new segments are mapped by the workers before any request that names them (same pipe, in order)."""
        out, fresh = [], []
        with self.lock:
            for seg in self.segments.get(slot_bytes, []):
                while seg.free and len(out) < count:
                    out.append((seg, seg.free.pop()))
            while len(out) < count:
                seg = self._new_segment(slot_bytes)
                fresh.append(seg)
                while seg.free and len(out) < count:
                    out.append((seg, seg.free.pop()))
        for seg in fresh:
            msg = b"A\t%d\t" % seg.sid + seg.path.encode("utf-8", "surrogateescape") + b"\n"
            for i in range(self.n):
                try:
                    self._send(i, msg)
                except DecodeError:                        # a worker that has ended maps nothing: its share of the un-link count is given up
                    with self.lock:
                        self.dead[i] = True
                        seg.unmapped_by -= 1
                        if seg.unmapped_by == 0:
                            seg.unlink()
        return out

    def _give_back(self, seg, slot):
        with self.lock:
            seg.free.append(slot)

    # ---- requests ----
    def submit(self, paths, slot_bytes):
        """One Ticket per file, in order.  The files go out in chunks of _CHUNK to the workers with the least outstanding work."""
        if self.closed:
            raise DecodeError("the decode farm is closed")
        paths = list(paths)
        for f in paths:
            if "\n" in f or "\x00" in f or "\r" in f:
                raise DecodeError("file names with line breaks cannot be handed to the decoder processes: %r" % f)
        slot_bytes = int(slot_bytes)
        with self.submit_lock:
            return self._submit(paths, slot_bytes)

    def _submit(self, paths, slot_bytes):
        places = self._take(slot_bytes, len(paths))
        tickets = []
        step = max(1, min(_CHUNK, -(-len(paths) // self.n)))
        for a in range(0, len(paths), step):
            group = list(range(a, min(a + step, len(paths))))
            by_seg = {}
            for j in group:                                # a chunk names one segment; a batch straddling two segments goes out as two chunks
                by_seg.setdefault(places[j][0].sid, []).append(j)
            for sid, js in by_seg.items():
                with self.lock:
                    alive = [w for w in range(self.n) if not self.dead[w]]
                    if not alive:
                        raise DecodeError("every decoder process has ended (exit codes %s)" % [p.poll() for p in self.procs])
                    i = min(alive, key=lambda w: self.load[w])
                    rid = self.next_id
                    self.next_id += 1
                    ch = _Chunk(i, len(js))
                    self.pending[rid] = ch
                    self.load[i] += len(js)
                seg = self.by_id[sid]
                items = b"\x00".join(b"%d\x00" % (places[j][1] * slot_bytes) + paths[j].encode("utf-8", "surrogateescape") for j in js)
                try:
                    self._send(i, b"D\t%d\t%d\t%d\t" % (rid, sid, slot_bytes) + items + b"\n")
                except DecodeError as e:
                    with self.lock:
                        self.pending.pop(rid, None)
                    ch.error = str(e)
                    ch.done.set()
                for k, j in enumerate(js):
                    tickets.append((j, Ticket(self, seg, places[j][1], ch, k, paths[j])))
        tickets.sort(key=lambda t: t[0])
        return [t for _, t in tickets]

    def close(self):
        if self.closed:
            return
        self.closed = True
        for p in self.procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                p.wait(timeout=5)
            except Exception:
                p.kill()
        with self.lock:
            for seg in self.by_id.values():
                seg.unlink()


_FARM = None
_FARM_LOCK = threading.Lock()


def farm_workers():
    """Decoder processes: ISX_DECODE_PROCS (0: none -- files are decoded on the thread pool of train/_common.py); default: the usable cores (at
    most 16) less one in eight, left to the thread that stacks, copies and launches (MI355X box, 16 cores: 12 workers 13.6 k images/s end to
    end, 16 workers 13.2 k)."""
    v = os.environ.get("ISX_DECODE_PROCS")
    if v is not None and v != "":
        return max(0, int(v))
    from utils.general import usable_cpus
    cores = min(16, usable_cpus())
    return max(1, cores - cores // 8)


def decode_farm():
    """The process-wide farm (started on first use), or None when ISX_DECODE_PROCS=0."""
    global _FARM
    n = farm_workers()
    if n <= 0:
        return None
    with _FARM_LOCK:
        if _FARM is not None and _FARM.pid != os.getpid():    # a forked child: the pipes belong to the parent
            _FARM = None
        if _FARM is None or _FARM.closed or _FARM.n != n:
            if _FARM is not None:
                _FARM.close()
            _FARM = DecodeFarm(n)
        return _FARM


def shutdown():
    global _FARM
    with _FARM_LOCK:
        if _FARM is not None:
            _FARM.close()
            _FARM = None


def decode_files(paths, slot_bytes, window=256, ahead=3):
    """[(H, W, 3) uint8 tensor of each file], file order kept, each tensor its own memory: the farm's eager form (queries, ragged folders).
    `ahead` windows of files are with the decoders while this thread copies the window in front out of its slots."""
    farm = decode_farm()
    paths = list(paths)
    out, inflight, nxt = [], [], 0
    try:
        while nxt < len(paths) or inflight:
            while nxt < len(paths) and len(inflight) < ahead:
                inflight.append(farm.submit(paths[nxt:nxt + window], slot_bytes))
                nxt += window
            tickets = inflight[0]
            for t in tickets:
                out.append(t.copy())
            for t in inflight.pop(0):
                t.release()
    finally:
        for tickets in inflight:
            for t in tickets:
                t.release()
    return out
