"""Option checkers, dataset-id / mean-std parsing and device helpers -- the subset of the
reference's utils/general.py that the retrieval entry points use (check_* :13-67,
parse_dataset_id :70-73, read_mean_std :76-80, move_device / tensor_t / tensor :94-106,
log / log_detail :137-146)."""
from __future__ import print_function

import os
import sys

import torch


def _fail(message, usage):
    print(message)
    usage()
    sys.exit(2)


def check_file(arg, name, should_exist, usage):
    present = os.path.isfile(arg)
    if should_exist and not present:
        _fail('Cannot find {0} file at path {1}\n'.format(name, arg), usage)
    if not should_exist and present:
        _fail('Cannot overwrite {0} file at path {1}\n'.format(name, arg), usage)
    return arg


def check_folder(arg, name, should_exist, usage):
    present = os.path.isdir(arg)
    if should_exist and not present:
        _fail('Cannot find {0} folder at path {1}\n'.format(name, arg), usage)
    if not should_exist and present:
        _fail('Cannot overwrite {0} folder at path {1}\n'.format(name, arg), usage)
    return arg


VALID_MODELS = ('alexnet', 'resnet152', 'resnet50')   # the reference admits the first two; resnet50 is BASELINE's choice


def check_model(arg, usage):
    if arg.lower() in VALID_MODELS:
        return arg.lower()
    _fail('Model {0} is not a valid model'.format(arg), usage)


def check_int(arg, name, usage):
    try:
        return int(arg)
    except ValueError:
        _fail('{0} was given as {1}. This is not an integer.\n'.format(name, arg), usage)


def check_bool(arg, name, usage):
    arg = arg.lower()
    if arg == '':
        _fail('{0} was not given. It should be a boolean (true/yes/y/1 for True and otherwise False).'.format(name), usage)
    return arg in ('true', 'yes', 'y', '1')


def parse_dataset_id(dataset_full):
    trimmed = dataset_full[:-1] if dataset_full.endswith('/') else dataset_full   # ONE trailing slash
    return trimmed.split('/')[-1]


def read_mean_std(fname):
    """Two lines of space-separated floats: per-channel mean, then std."""
    with open(fname) as f:
        mean = [float(v) for v in f.readline().split(' ')]
        std = [float(v) for v in f.readline().split(' ')]
    return mean, std


def move_device(obj, device):
    """device >= 0 -> the current GPU (the caller selects it with torch.cuda.device), else CPU."""
    return obj.cuda() if device >= 0 else obj.cpu()


def tensor_t(t, device, *sizes):
    return move_device(t(*sizes), device)


def tensor(device, *sizes):
    """Uninitialised fp32 tensor, allocated directly where it will live."""
    return torch.empty(*sizes, dtype=torch.float32, device='cuda' if device >= 0 else 'cpu')


def usable_cpus():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container can see 256 CPUs and own 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:                                                     # cgroup v2
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except Exception:
        try:                                                 # cgroup v1
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0 and period > 0:
                n = min(n, max(1, int(float(quota) / period + 0.5)))
        except Exception:
            pass
    # one rank per GPU on a node (torch.distributed.run sets LOCAL_WORLD_SIZE): the ranks share the container's CPUs
    try:
        n = max(1, n // max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1'))))
    except ValueError:
        pass
    return max(1, n)


def cap_torch_threads():
    """torch sizes its OpenMP team from the CPUs it SEES; under a cgroup quota (16 of 256 on the MI355X boxes) a team of 128 is throttled into
    the ground -- elementwise host ops at 6 GB/s, every host-side pass of an evaluation run several times slower.  The entry points of this
    package cap the team at what the process owns; a smaller explicit setting (OMP_NUM_THREADS, torch.set_num_threads) is left alone."""
    n = usable_cpus()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def is_main_process():
    """True on rank 0 of a torch.distributed job and in a plain single-process run: the one process that prints result
    lines, appends to the log file and writes checkpoints."""
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0


def log(P, message):
    if not is_main_process():
        return
    print(message)
    log_file = getattr(P, 'log_file', None)
    if log_file:
        with open(log_file, 'a') as f:
            f.write(str(message) + '\n')


def log_detail(P, message, detail):
    if message is not None:
        log(P, message)
    log_file = getattr(P, 'log_file', None)
    if log_file and detail and is_main_process():
        with open(log_file, 'a') as f:
            f.write(str(detail) + '\n')
