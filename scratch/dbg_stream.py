import sys, numpy as np, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
M = 16384
for Cout in (64, 256):
    x = (np.arange(M, dtype=np.float32)[:, None] * 64 + np.arange(64, dtype=np.float32)[None, :]).astype(np.float32)
    w = np.zeros((Cout, 64), np.float32)
    for c in range(Cout): w[c, c % 64] = 1.0
    b = np.zeros(Cout, np.float32)
    xt = torch.from_numpy(x).cuda().view(1, M, 1, 64).permute(0, 3, 1, 2)
    y = ops.conv1x1_nhwc(xt, torch.from_numpy(w).cuda(), torch.from_numpy(b).cuda(), None, False).permute(0, 2, 3, 1).reshape(M, Cout).cpu().numpy()
    want = x[:, np.arange(Cout) % 64]
    bad = y != want
    print("Cout", Cout, "mismatch", bad.sum(), "of", bad.size)
    if bad.any():
        ps, cs = np.nonzero(bad)
        print(" bad channels (mod 64):", sorted(set((cs % 64).tolist()))[:64])
        print(" bad pixels mod 128:", sorted(set((ps % 128).tolist()))[:130])
        for p, c in list(zip(ps, cs))[:12]:
            g = y[p, c]; print("  y[%d][%d] = %g -> pixel %d, k %d (want pixel %d k %d)" % (p, c, g, int(g) // 64, int(g) % 64, p, c % 64))
