import os, sys, random, time
sys.path.insert(0, "/root/repo/instance-search_amd")
import torch
from torch.profiler import profile, ProfilerActivity
from train import siamese_descriptor as sd
from utils.dataset import synthetic_image_set
torch.manual_seed(0); random.seed(0)
P = sd.P
P.cuda_device, P.cnn_model, P.feature_size2d, P.feature_dim = 0, "resnet50", (7, 7), 2048
P.train_epochs, P.train_batch_size, P.train_micro_batch, P.test_batch_size = 1, 64, 8, 128
P.train_loss_int, P.train_test_int, P.untrained_blocks, P.train_epoch_switch = 10 ** 9, 10 ** 9, -1, 1
tr = synthetic_image_set(128, 16, seed=1)
te = synthetic_image_set(32, 16, seed=2)
sd.test_print_descriptor = lambda *a, **k: 0
sd.main(tr, tr, te)                      # warm-up epoch (MIOpen find etc.)
torch.cuda.synchronize()
t = time.perf_counter()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    sd.main(tr, tr, te)
    torch.cuda.synchronize()
print("epoch wall", time.perf_counter() - t)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=60))
