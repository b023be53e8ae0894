#!/bin/bash
# Smoke of the training command line on the GPU: a synthetic spec, then a folder of JPEG files (decoded by the decoder processes, resident in HBM).
set -e
cd "$(dirname "$0")/../instance-search_amd"
python -m train.siamese_descriptor --dataset=synthetic:CLICIDE_video_224sq:n=64:q=16:labels=8 --model=resnet50 --device=0 --epochs=1 --batch-size=16 --micro-batch=4 --feature-dim=256 | tail -4
TMP=$(mktemp -d)
python - "$TMP" <<'PY'
import os, sys, numpy as np
from PIL import Image
root = os.path.join(sys.argv[1], "CLICIDE_video_224sq")
os.makedirs(os.path.join(root, "test")); os.makedirs(os.path.join(sys.argv[1], "data"))
open(os.path.join(sys.argv[1], "data", "CLICIDE_224sq_train_ms.txt"), "w").write("0.485 0.456 0.406\n0.229 0.224 0.225\n")
for i in range(96):
    a = np.random.default_rng(i).integers(0, 256, (224, 224, 3), dtype=np.uint8)
    Image.fromarray(a).save(os.path.join(root, "l%02d-%d.jpg" % (i % 8, i // 8)), quality=90)
for i in range(16):
    a = np.random.default_rng(1000 + i).integers(0, 256, (224, 224, 3), dtype=np.uint8)
    Image.fromarray(a).save(os.path.join(root, "test", "l%02d-q%d.jpg" % (i % 8, i // 8)), quality=90)
PY
REPO=$(pwd)
(cd "$TMP" && PYTHONPATH="$REPO" python -m train.siamese_descriptor --dataset="$TMP/CLICIDE_video_224sq" --model=resnet50 --device=0 --epochs=1 --batch-size=16 --micro-batch=4 --feature-dim=256 | tail -4)
rm -rf "$TMP"
