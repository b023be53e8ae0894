"""Where does a workgroup of the 1x1-convolution GEMM spend its life?  Needs a STAMPS build of libisx:
    bash tools/build_variant.sh stamps -DISX_STAMPS=1 && ISX_LIB=build_ab/stamps/libisx.so python tools/conv_phase_lab.py
Wave 0 of every workgroup records the shader clock (s_memtime) at: 0 entry, 1 first k-tile staged, 2 main loop done, 3 residual landed (the lab build
drains the loads there), 4 stores issued, 5 stores complete.  Prints, per layer shape: kernel ms, the mean length of each phase over the steady-state
workgroups in microseconds, and how many workgroups were alive on average (sum of lifetimes / kernel span)."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "instance-search_amd"))
import torch  # noqa: E402
from isx import ops  # noqa: E402
from isx._lib import LIB_PATH  # noqa: E402

raw = ctypes.CDLL(LIB_PATH)
if not hasattr(raw, "isx_debug_set_stamps"):
    raise SystemExit("conv_phase_lab: %s is not a STAMPS build (tools/build_variant.sh stamps -DISX_STAMPS=1)" % LIB_PATH)
raw.isx_debug_set_stamps.argtypes = [ctypes.c_void_p]
B = int(os.environ.get("LAB_B", "1024"))
cl = lambda t: t.contiguous(memory_format=torch.channels_last)
shapes = [(28, 128, 512, True), (28, 512, 128, False), (14, 256, 1024, True), (14, 1024, 256, False), (7, 512, 2048, True), (56, 256, 128, False)]
for H, Cin, Cout, res in shapes:
    x = cl(torch.relu(torch.randn(B, Cin, H, H, device="cuda")))
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") * Cin ** -0.5
    b = torch.randn(Cout, device="cuda")
    r = cl(torch.randn(B, Cout, H, H, device="cuda")) if res else None
    M = B * H * H
    nwg = ((M + 127) // 128) * ((Cout + 127) // 128) + 4096            # upper bound incl. the 64x64 tail tiles of the same grid
    st = torch.zeros((nwg, 8), dtype=torch.int64, device="cuda")
    for _ in range(3):
        ops.conv1x1_nhwc(x, w, b, r, True)
    torch.cuda.synchronize()
    raw.isx_debug_set_stamps(ctypes.c_void_p(st.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.conv1x1_nhwc(x, w, b, r, True); e1.record()
    torch.cuda.synchronize()
    raw.isx_debug_set_stamps(None)
    ms = e0.elapsed_time(e1)
    h = st.cpu().numpy()
    h = h[h[:, 0] > 0]
    # s_memtime is a per-XCD / clock-gated shader-cycle counter: only differences INSIDE one workgroup mean anything (epochs differ by seconds between
    # XCDs and drift inside one).  Steady state = the middle three quarters of the workgroups in block order.
    mid = h[len(h) // 8: len(h) - len(h) // 8]
    cols = [1, 2, 3, 4, 5] if res else [1, 2, 4, 5]
    names = {1: "prologue", 2: "main loop", 3: "residual round trip", 4: "add + issue stores", 5: "stores drain"}
    life = float((mid[:, 5] - mid[:, 0]).mean())
    prev, parts = 0, []
    for c in cols:
        d = float((mid[:, c] - mid[:, prev]).mean())
        parts.append("%s %.0f (%.0f %%)" % (names[c], d, 100.0 * d / life))
        prev = c
    slots = min(len(h), 512)
    implied = life * len(h) / slots / (ms * 1e3)                       # cycles per microsecond if `slots` workgroups were resident throughout = the clock in MHz
    flop = 2.0 * M * Cin * Cout
    print("1x1 H=%3d %5d->%5d res=%d: %.3f ms %.1f TF | %d workgroups | per workgroup, shader cycles: %s | life %.0f cycles (x %d / %d slots / kernel time = %.2f GHz)" %
          (H, Cin, Cout, res, ms, flop / ms / 1e9, len(h), ", ".join(parts), life, len(h), slots, implied / 1e3), flush=True)
