"""Who carries the GPU-vs-CPU descriptor difference?  fp64 evaluation of the same net as arbiter (torch, GPU, MIOpen off)."""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd"); sys.path.insert(0, "/root/repo/tests")
from isx import backbones, ops
from model.siamese import TuneClassif
from model.nn_utils import fold_batch_norm
from utils.dataset import synthetic_image_set
import test_gpu_end_to_end as T
for arch in ("resnet50", "resnet152"):
    w = T._calibrated_weights("classif", 10, "/tmp/w_%s.pth" % arch, arch=arch)
    net = TuneClassif(backbones.MODELS[arch](pretrained=True), 10)
    net.load_state_dict(torch.load(w)); net.eval()
    x = torch.stack([t for t, _, _ in synthetic_image_set(80, 10, seed=1234, structure=0.7)])
    def desc(f): 
        p = f.mean((2, 3)); return p / (p.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()
    with torch.no_grad():
        d_cpu = desc(net.features(x))
        torch.backends.cudnn.enabled = False
        d64 = desc(net.double().cuda().features(x.double().cuda())).cpu()
        torch.backends.cudnn.enabled = True
        net = net.float().cpu()
        f = fold_batch_norm(net.features).cuda().to(memory_format=torch.channels_last)
        d_gpu = ops.gap_l2(f(x.cuda())).cpu()
        d_gpu_nofold = desc(net.cuda().features(x.cuda())).cpu()
    q, g = slice(0, 20), slice(20, 80)
    cos = lambda d: d[q].double() @ d[g].double().t()
    for name, d in (("gpu isx folded", d_gpu), ("cpu torch fp32", d_cpu), ("gpu torch/MIOpen fp32 unfolded", d_gpu_nofold)):
        print(arch, name, "max|ddesc vs f64| %.3g  max|dcos vs f64| %.3g" % (float((d.double() - d64).abs().max()), float((cos(d) - cos(d64)).abs().max())))
    print(arch, "gpu vs cpu: ddesc %.3g dcos %.3g" % (float((d_gpu - d_cpu).abs().max()), float((cos(d_gpu) - cos(d_cpu)).abs().max())))
