import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
dev = "cuda"
def unit(n, d, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = torch.randn(n, d, device=dev, generator=g)
    return x / x.norm(dim=1, keepdim=True)
M, N, D, k = 10000, 125000, 2048, 100
Q, G = unit(M, D, 20), unit(N, D, 21)
gh = ops.gallery_to_f16(G)
ref = ops.cosine_topk_fast(Q, G, k, gallery_f16=gh)
def run_split(parts):
    bounds = [0]
    step = ((M + parts - 1) // parts + 255) // 256 * 256
    while bounds[-1] < M: bounds.append(min(M, bounds[-1] + step))
    streams = [torch.cuda.Stream() for _ in range(len(bounds) - 1)]
    wss = [torch.empty((ops.cosine_topk_fast_workspace(bounds[i + 1] - bounds[i], N, D, k, True),), device=dev, dtype=torch.uint8) for i in range(len(streams))]
    ts = torch.empty((M, k), device=dev); ti = torch.empty((M, k), device=dev, dtype=torch.int64)
    def go():
        cur = torch.cuda.current_stream()
        for i, st in enumerate(streams):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                a, b = bounds[i], bounds[i + 1]
                ops.cosine_topk_fast(Q[a:b], G, k, gallery_f16=gh, ws=wss[i], out=(ts[a:b], ti[a:b]))
        for st in streams: cur.wait_stream(st)
    go(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5): go()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 5
    ok = torch.equal(ti, ref[1]) and torch.equal(ts, ref[0])
    print(f"{parts} stream(s): {dt*1e3:.2f} ms  {M*N/dt/1e9:.1f} G dist/s equal={ok}", flush=True)
for p in (1, 2, 3, 4, 2, 1):
    run_split(p)
