import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "instance-search_amd"))
import torch
from isx import backbones
from model.siamese import TuneClassifSub
from train._common import prepare_for_inference
from train.classif_regions import P, _best_location_descriptors
torch.manual_seed(0)
for n_cls in (3, 464):
    m = TuneClassifSub(backbones.resnet50(pretrained=True, seed=0), n_cls, (7, 7)).cuda()
    P.fold_bn = True
    prepare_for_inference(m, P)
    x = torch.randn(7, 3, 288, 288, device="cuda").contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        whole = m(x)[0]
        parts = torch.cat([m(x[i:i + 1])[0] for i in range(7)], 0)
        two = torch.cat([m(x[:3])[0], m(x[3:])[0]], 0)
        f_w = m.features(x); f_p = torch.cat([m.features(x[i:i+1]) for i in range(7)], 0)
        r_w = m.feature_reduc(f_w); r_p = torch.cat([m.feature_reduc(f_p[i:i+1]) for i in range(7)], 0)
    print(n_cls, "features", torch.equal(f_w, f_p), "reduc", torch.equal(r_w, r_p), "scores 1-by-1", torch.equal(whole, parts), float((whole - parts).abs().max()),
          "3+4", torch.equal(whole, two), "desc", torch.equal(_best_location_descriptors(whole), _best_location_descriptors(parts)))
