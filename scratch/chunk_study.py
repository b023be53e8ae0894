"""Round 5 study (CPU only): how far from a float64 evaluation does the BN-folded ResNet trunk land when every convolution output is
(a) ONE k-ordered fp32 fma chain (rounds 1-4), (b) chains over chunks of the reduction folded into a second accumulator, against
(c) torch's CPU fp32 path (the reference's path).  Decides the chunk length of the round-5 kernels.
usage: python scratch/chunk_study.py resnet50|resnet152 [n_images]"""
import ctypes, os, subprocess, sys, time
import numpy as np
import torch

R = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(R, "..", "instance-search_amd"))
so = os.path.join(R, "chunk_study.so")
if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(R, "chunk_study.c")):
    subprocess.check_call(["gcc", "-O3", "-march=native", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", os.path.join(R, "chunk_study.c"), "-o", so])
L = ctypes.CDLL(so)
fp = ctypes.POINTER(ctypes.c_float)
L.conv_chunked.argtypes = [fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                           ctypes.c_int, fp, fp, ctypes.c_int, fp]


def P(a):
    return a.ctypes.data_as(fp) if a is not None else None


def conv(x, w_oihw, bias, stride, pad, chunk, res=None, relu=True):
    """x (B,H,W,Cin) numpy; weight (Cout,Cin,KH,KW) torch"""
    co, ci, kh, kw = w_oihw.shape
    wT = np.ascontiguousarray(w_oihw.permute(2, 3, 1, 0).reshape(kh * kw * ci, co).numpy())
    B, H, W, _ = x.shape
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    y = np.empty((B, Ho, Wo, co), np.float32)
    b = np.ascontiguousarray(bias.numpy())
    L.conv_chunked(P(x), B, H, W, ci, P(wT), co, kh, kw, stride, pad, chunk(kh * kw * ci, ci) if callable(chunk) else chunk, P(b), P(res), int(relu), P(y))
    return y


def maxpool(y):
    t = torch.from_numpy(y).permute(0, 3, 1, 2)
    return np.ascontiguousarray(torch.nn.functional.max_pool2d(t, 3, 2, 1).permute(0, 2, 3, 1).numpy())


def run_trunk(folded, x_nchw, chunk):
    """folded: fold_batch_norm(features) (CPU modules used only as weight containers)"""
    from model.nn_utils import _StemConvPool, _FusedBlock, _ChannelsLastEntry
    x = np.ascontiguousarray(x_nchw.permute(0, 2, 3, 1).numpy())
    for m in folded:
        if isinstance(m, _ChannelsLastEntry):
            continue
        if isinstance(m, _StemConvPool):
            c = m.cba
            x = maxpool(conv(x, c.conv.weight, c.bias, 2, 3, chunk))
        elif isinstance(m, _FusedBlock):
            t = x
            for c in m.convs[:-1]:
                t = conv(t, c.conv.weight, c.bias, c.conv.stride[0], c.conv.padding[0], chunk)
            last = m.convs[-1]
            if m.downsample is not None:
                s = m.downsample.conv.stride[0]
                cat = np.ascontiguousarray(np.concatenate([t, x[:, ::s, ::s, :]], axis=3))
                w = torch.cat([last.conv.weight, m.downsample.conv.weight], 1)
                x = conv(cat, w, last.bias, 1, 0, chunk)
            else:
                x = conv(t, last.conv.weight, last.bias, 1, 0, chunk, res=x)
        else:
            raise RuntimeError(type(m))
    return x


def main():
    arch = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    from isx import backbones
    from model.siamese import TuneClassif
    from model.nn_utils import fold_batch_norm
    from utils.dataset import synthetic_image_set
    torch.manual_seed(0)
    net = TuneClassif(backbones.MODELS[arch](pretrained=True), 10)
    cal = torch.stack([t for t, _, _ in synthetic_image_set(32, 10, seed=99, structure=0.7)])
    net.train()
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.momentum = None
            m.reset_running_stats()
    with torch.no_grad():
        net.features(cal)
    net.eval()
    x = torch.stack([t for t, _, _ in synthetic_image_set(n, 10, seed=1234, structure=0.7)])

    def desc(f):
        p = f.mean((2, 3))
        return p / (p.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()

    def desc_nhwc(f):
        p = torch.from_numpy(f).permute(0, 3, 1, 2).mean((2, 3))       # fp32 pooling as the GPU path does
        return p / (p.pow(2).sum(1, keepdim=True) + 1e-10).sqrt()

    with torch.no_grad():
        t0 = time.time(); d_cpu = desc(net.features(x)); t1 = time.time()
        d64 = desc(net.double().features(x.double())); t2 = time.time()
        net.float()
        folded = fold_batch_norm(net.features)
        d_fold_cpu = desc(folded(x))
    print("torch fp32 %.1fs, fp64 %.1fs" % (t1 - t0, t2 - t1), flush=True)
    q = slice(0, n // 4); g = slice(n // 4, n)
    cos = lambda d: d[q].double() @ d[g].double().t()
    e = lambda d: float((cos(d) - cos(d64)).abs().max())
    er = lambda d: float((cos(d) - cos(d64)).pow(2).mean().sqrt())
    print("%s: torch-CPU fp32 unfolded: max %.3g rms %.3g | folded (torch CPU): max %.3g rms %.3g" % (arch, e(d_cpu), er(d_cpu), e(d_fold_cpu), er(d_fold_cpu)), flush=True)
    policies = [("64", 64), ("64, stem one chain", lambda K, ci: 0 if K == 147 else 64), ("64, stem 63 (3 filter rows)", lambda K, ci: 63 if K == 147 else 64),
                ("32", 32), ("64, K=64..256 one chain", lambda K, ci: 64 if K > 256 or K == 147 else 0)]
    for name, ch in policies:
        t0 = time.time()
        d = desc_nhwc(run_trunk(folded, x, ch))
        print("%s: chunk %-22s max|cos-cos64| %.3g rms %.3g  ratio to torch-CPU %.2f (rms %.2f)  [%.0fs]"
              % (arch, name, e(d), er(d), e(d) / e(d_cpu), er(d) / er(d_cpu), time.time() - t0), flush=True)


main()
