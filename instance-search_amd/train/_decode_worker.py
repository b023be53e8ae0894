"""Decoder process of train/_decode_farm.py.  Stand-alone on purpose: imports the standard library, numpy and PIL only (never torch, never the
GPU), is started with its file name (`python _decode_worker.py`), reads requests on stdin and answers on stdout; ends when stdin closes (the
parent went away).

    A\t<segment id>\t<path of the shared file>\n                       map a segment                      -> A\t<segment id>\n
    D\t<request id>\t<segment id>\t<slot bytes>\t<off>\x00<file>\x00<off>\x00<file>...\n     decode the files, each into its slot
                                                                        -> D\t<request id>\t<k|b|e> <h> <w> [message];...\n
        k: decoded, (h, w, 3) uint8 at the slot; b: does not fit the slot (h, w reported, nothing written); e: the decoder raised (message).

What is decoded is what the reference reads (utils/image.py:211-214, imread_rgb): the file as 8-bit RGB, rows top to bottom."""
import mmap
import os
import sys


def main():
    import numpy as np
    from PIL import Image
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    segs = {}
    for line in inp:
        line = line.rstrip(b"\n")
        if not line:
            continue
        kind, _, rest = line.partition(b"\t")
        if kind == b"A":
            sid, _, path = rest.partition(b"\t")
            fd = os.open(path.decode("utf-8", "surrogateescape"), os.O_RDWR)
            try:
                mm = mmap.mmap(fd, 0)
            finally:
                os.close(fd)
            segs[sid] = np.frombuffer(mm, dtype=np.uint8)
            out.write(b"A\t" + sid + b"\n")
            out.flush()
            continue
        rid, sid, cap, items = rest.split(b"\t", 3)
        buf, cap = segs[sid], int(cap)
        parts = items.split(b"\x00")
        res = []
        for k in range(0, len(parts) - 1, 2):
            off, fname = int(parts[k]), parts[k + 1].decode("utf-8", "surrogateescape")
            try:
                a = np.asarray(Image.open(fname).convert("RGB"), dtype=np.uint8)
                h, w = a.shape[0], a.shape[1]
                if a.size <= cap:
                    buf[off:off + a.size] = a.reshape(-1)
                    res.append(b"k %d %d" % (h, w))
                else:
                    res.append(b"b %d %d" % (h, w))
            except Exception as e:                       # reported to the caller, which raises it where the image is asked for
                msg = ("%s: %s" % (type(e).__name__, e)).replace("\n", " ").replace(";", ",").replace("\t", " ")
                res.append(b"e 0 0 " + msg.encode("utf-8", "replace"))
        out.write(b"D\t" + rid + b"\t" + b";".join(res) + b"\n")
        out.flush()


if __name__ == "__main__":
    main()
