"""Does running the trunk as two half-batches on two HIP streams (kernel tails and launch gaps of one filled by the other) beat one batch of 1024?"""
import sys, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
net = bench.build_net("resnet50", "f32", dev, channels_last=True, fold_bn=True)
x = torch.randn(1024, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
def t(f, n=6, w=2):
    for _ in range(w): f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
with torch.no_grad():
    one = t(lambda: net.features(x))
    print("one stream, 1024 images: %.2f ms" % one, flush=True)
    for parts in (2, 4):
        streams = [torch.cuda.Stream() for _ in range(parts)]
        S = 1024 // parts
        def run():
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(cur)
            outs = []
            for i, s in enumerate(streams):
                s.wait_event(ev)
                with torch.cuda.stream(s):
                    outs.append(net.features(x[i * S:(i + 1) * S]))
            for s in streams:
                e = torch.cuda.Event(); e.record(s); cur.wait_event(e)
            return outs
        ms = t(run)
        ref = net.features(x)
        same = torch.equal(torch.cat(run(), 0), ref)
        print("%d streams x %d images: %.2f ms   identical %s" % (parts, S, ms, same), flush=True)
