"""Host-side checks of the measurement tools (no GPU): the pieces bench.py borrows from tools/ fit together."""
import inspect
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_builds_the_training_leg_arguments_the_tool_reads():
    """bench.py's training leg (bench_legs/training.py) calls tools/bench_train.run_config with the tool's own parser defaults: every `args.<name>` run_config reads
    exists on them (a hand-written Namespace once missed a new flag and the leg reported an AttributeError instead of a figure)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_train
    args = bench_train.make_parser().parse_args(["--images", "512", "--labels", "64", "--epochs", "5", "--backbone", "resnet50"])
    used = set(re.findall(r"\bargs\.([a-z_]+)", inspect.getsource(bench_train.run_config)))
    assert used and all(hasattr(args, u) for u in used), sorted(u for u in used if not hasattr(args, u))
    src = open(os.path.join(ROOT, "bench_legs", "training.py")).read()          # the training leg of bench.py
    assert "bench_train.make_parser().parse_args(" in src and "Namespace(images=" not in src
    assert "leg_training.measure" in open(os.path.join(ROOT, "bench.py")).read()
