#!/bin/bash
# The rocprofv3 passes behind profiles/<tag>_* (run on the GPU box: gpurun -- bash tools/profile_round.sh r03).  Counter passes are runs of their
# own with --kernel-trace only (gpurun refuses --pmc combined with the sys / hip / hsa trace domains).
set -e
TAG=${1:-r06}
OUT=gpurun_out/$TAG
REPO=$(pwd)
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $REPO
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_line.err
cp gpurun_out/bench_detail.json $OUT/bench_detail_full.json
echo "bench line done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-regions-bench --ingest-images 0 --decode-images 0 --no-train-bench > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/regions -- python3 bench.py --only-regions --no-train-bench > $OUT/regions.log 2>&1
echo "regions trace done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard-bench --no-kernel-pass --no-regions-bench --ingest-images 0 --decode-images 0 --no-train-bench > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard-bench --no-kernel-pass --no-regions-bench --ingest-images 0 --decode-images 0 --no-train-bench > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/mfma -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-shard-bench --no-kernel-pass --no-regions-bench --ingest-images 0 --decode-images 0 --no-train-bench > $OUT/mfma.log 2>&1
echo "mfma done"
python3 profiles/summarize_prof.py $TAG $OUT/stats $OUT/fetch $OUT/write $OUT/mfma > $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/rfetch -- python3 bench.py --only-regions --no-train-bench > $OUT/rfetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/rwrite -- python3 bench.py --only-regions --no-train-bench > $OUT/rwrite.log 2>&1
python3 profiles/summarize_regions_pmc.py $TAG $OUT/rfetch $OUT/rwrite 128 >> $OUT/summarize.log 2>&1 || tail -5 $OUT/summarize.log
echo "regions pmc done"
mkdir -p $OUT/summaries && cp profiles/${TAG}_* profiles/roofline_traffic.json $OUT/summaries/ 2>/dev/null || true
# the regions leg: per-kernel table of its own trace
python3 - <<PY
import csv, glob, collections
f = sorted(glob.glob("$OUT/regions/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1]))) if f else []
with open("$OUT/summaries/${TAG}_regions_kernel_stats.csv", "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for r in rows:
        w.writerow([r.get("Name"), r.get("Calls"), float(r.get("TotalDurationNs", 0)) / 1e3, float(r.get("AverageNs", 0)) / 1e3, r.get("Percentage")])
PY
# siamese training on the reference's configuration (layer4 trained): which kernels run in the step (tools/bench_train.py)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 tools/bench_train.py --configs reference --epochs 3 > $OUT/train.log 2>&1
mkdir -p $OUT/summaries
TRACE=$(ls $OUT/train/*/*kernel_trace.csv $OUT/train/*kernel_trace.csv 2>/dev/null | head -1)
[ -n "$TRACE" ] && python3 tools/train_step_trace.py $TRACE $OUT/summaries/${TAG}_train_step_launches.csv || true      # one steady step's launches, in order
rm -f $OUT/train/*/*kernel_trace.csv $OUT/train/*kernel_trace.csv
python3 - <<PY
import csv, glob
f = sorted(glob.glob("$OUT/train/**/*kernel_stats.csv", recursive=True))
rows = list(csv.DictReader(open(f[-1]))) if f else []
with open("$OUT/summaries/${TAG}_train_kernel_stats.csv", "w") as o:
    w = csv.writer(o); w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
    for r in rows[:60]:
        w.writerow([r.get("Name")[:160], r.get("Calls"), float(r.get("TotalDurationNs", 0)) / 1e3, float(r.get("AverageNs", 0)) / 1e3, r.get("Percentage")])
PY
grep '^{' $OUT/train.log | tail -1 > $OUT/summaries/${TAG}_bench_train.json
cp $OUT/bench_detail_full.json $OUT/summaries/${TAG}_bench_detail.json 2>/dev/null || true
tail -1 $OUT/bench_line.json > $OUT/summaries/${TAG}_bench_line.json
rm -rf $OUT/stats $OUT/regions $OUT/fetch $OUT/write $OUT/mfma $OUT/rfetch $OUT/rwrite $OUT/train
ls $OUT/summaries
