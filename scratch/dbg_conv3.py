import sys, numpy as np, torch
sys.path.insert(0, "/root/repo/instance-search_amd"); sys.path.insert(0, "/root/repo/oracle")
from isx import ops
import oracle as O
B, H, W, Cin, Cout, s = 1, 4, 4, 32, 32, 1
x = np.zeros((B, H, W, Cin), np.float32)
for h in range(H):
    for w_ in range(W): x[0, h, w_, :] = 100 * h + 10 * w_ + np.arange(Cin) * 0.01
w = np.zeros((Cout, 3, 3, Cin), np.float32)
for c in range(Cout): w[c, c % 3, (c // 3) % 3, c % Cin] = 1.0          # output c picks tap (c%3, (c//3)%3), channel c
b = np.zeros(Cout, np.float32)
want = O.conv3x3_nhwc(x, w, b, s, None, False)
xt = torch.from_numpy(x).cuda().permute(0, 3, 1, 2)
y = ops.conv3x3_nhwc(xt, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), s, None, False).permute(0, 2, 3, 1).cpu().numpy()
bad = y != want
print("mismatch", bad.sum(), "of", bad.size)
idx = np.argwhere(bad)
for (bb, h, w_, c) in idx[:24]:
    print("pixel (%d,%d) cout %2d tap (%d,%d): got %8.2f want %8.2f" % (h, w_, c, c % 3, (c // 3) % 3, y[bb, h, w_, c], want[bb, h, w_, c]))
