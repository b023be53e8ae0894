"""get_embeddings throughput through the drop-in path at the reference's batch size, with and without the device batching."""
import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import backbones
from model.nn_utils import fold_batch_norm
from model.siamese import TuneClassif
from train import _common as TC
from train import classif_finetune as cf
from utils.dataset import synthetic_images
torch.manual_seed(0)
net = TuneClassif(backbones.resnet50(pretrained=True, seed=0), 5).eval()
net.features = fold_batch_norm(net.features)
net = net.cuda().to(memory_format=torch.channels_last)
imgs = synthetic_images(64, seed=3)
N = 4096
data = [(imgs[i % 64], "l%d" % (i % 5), "p%d" % i) for i in range(N)]
P = cf.P
P.cuda_device, P.embeddings_classify, P.test_pre_proc = 0, False, True
for pixels in (0, 512 * 224 * 224):
    TC._MIN_DEVICE_BATCH_PIXELS = pixels
    for bs in (64, 256):
        P.test_batch_size = bs
        cf.get_embeddings(net, data, 0, 2048); torch.cuda.synchronize()
        t = time.perf_counter(); cf.get_embeddings(net, data, 0, 2048); torch.cuda.synchronize(); dt = time.perf_counter() - t
        print(f"device batching {'on ' if pixels else 'off'} --batch={bs}: {N/dt:.0f} images/s", flush=True)
