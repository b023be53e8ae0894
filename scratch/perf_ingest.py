import sys, time, torch
sys.path.insert(0, "/root/repo/instance-search_amd")
from isx import ops
B = 1024
u8 = torch.randint(0, 256, (B, 224, 224, 3), dtype=torch.uint8).pin_memory()
f32 = torch.randn(B, 3, 224, 224).pin_memory()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
d8 = u8.cuda()
t_k = timeit(lambda: ops.images_u8_to_f32(d8, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)))
t_h8 = timeit(lambda: u8.cuda(non_blocking=True))
t_h32 = timeit(lambda: f32.cuda(non_blocking=True))
print(f"B={B}: isx_images_u8_to_f32 {t_k:.3f} ms ({B*224*224*15/t_k/1e6:.0f} GB/s) | H2D uint8 {t_h8:.2f} ms ({u8.numel()/t_h8/1e6:.1f} GB/s) | H2D fp32 {t_h32:.2f} ms ({f32.numel()*4/t_h32/1e6:.1f} GB/s)")
