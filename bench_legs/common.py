"""Shared by bench.py and its side legs: the peaks the rooflines are priced against, the trunk under test, host facts, the PMC traffic profile."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.path.join(ROOT, "instance-search_amd") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))

PEAK_F16_MFMA_TFLOPS = 2500.0    # dense fp16/bf16 MFMA peak of one MI355X (MI355X_MICROARCH.md)
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0            # HBM3E spec
RESNET50_GFLOP_PER_IMAGE = 8.17  # 2 x 4.087 GMAC, convolutions of the 224x224 trunk (~8.2)


def build_net(name, dtype, device, channels_last=False, fold_bn=False):
    import torch
    from isx import backbones
    from model.nn_utils import set_net_train
    from model.siamese import TuneClassif
    torch.manual_seed(0)
    net = TuneClassif(backbones.MODELS[name](pretrained=True, seed=0), 464)
    set_net_train(net, False)
    if fold_bn:
        from model.nn_utils import fold_batch_norm
        net.features = fold_batch_norm(net.features)
    net = net.to(device)
    if dtype == "bf16" or channels_last:
        net = net.to(memory_format=torch.channels_last)
    return net


def usable_cpus():
    """CPUs this process may really use: affinity mask, capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def csrc_digest():
    """sha256 (16 hex digits) over the kernel sources (instance-search_amd/csrc/*.{hip,hpp,cpp}, Makefile, include/isx.h): what a PMC profile is a
    profile OF.  profiles/summarize_prof.py stamps it into roofline_traffic.json; a bench run whose sources hash differently reports the
    traffic as null (`traffic_stale`) instead of bytes that belong to other kernels.  (The GPU box has no .git: a content hash, not a commit.)"""
    import glob
    import hashlib
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "instance-search_amd", "csrc")
    for path in sorted(glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")) + glob.glob(os.path.join(csrc, "*.cpp"))
                       + [os.path.join(csrc, "Makefile"), os.path.join(ROOT, "include", "isx.h")]):
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def load_traffic():
    """HBM bytes measured with rocprofv3 PMC passes on an EARLIER run of this command (profiles/roofline_traffic.json, written
    by profiles/summarize_prof.py): a property of that profiled run, stamped with its source -- never of the run printing it."""
    path = os.path.join(ROOT, "profiles", "roofline_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return {}
    try:
        t["fresh"] = bool(t.get("csrc_digest")) and t.get("csrc_digest") == csrc_digest()
    except Exception:
        t["fresh"] = False
    if not t["fresh"]:                               # the kernels changed since the counters were read: no bytes rather than stale bytes
        t = {"source": t.get("source"), "csrc_digest": t.get("csrc_digest"), "fresh": False, "kernels": {}, "regions_leg": {}}
    return t
