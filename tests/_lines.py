"""The result lines an evaluation main prints (reference test/*_test.py), picked out of a run's stdout.  Under a multi-rank gloo launch the
transport's own "[Gloo] Rank r is connected ..." chatter shares that stdout and may land in front of a line on the same row."""
import re

_LINE = re.compile(r"(Classification|Descriptor|Testing|Loading).*")


def printed_lines(out):
    got = []
    for row in out.splitlines():
        m = _LINE.search(row)
        if m and "[Gloo]" not in m.group(0):
            got.append(m.group(0))
    return got
