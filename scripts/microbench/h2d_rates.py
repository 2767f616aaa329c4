"""Host -> device feed rates on this box (GPU only): pageable->pinned memcpy, pinned->device DMA, pageable->device."""
import time, numpy as np, torch
dev = torch.device("cuda:0")
n = 256 * 16 * 2048
src = torch.from_numpy(np.random.default_rng(0).standard_normal(n).astype(np.float32))
pin = torch.empty(n, dtype=torch.float32).pin_memory()
dst = torch.empty(n, dtype=torch.float32, device=dev)
mb = n * 4 / 1e6
def rate(f, reps=10, sync=False):
    f()
    if sync: torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): f()
    if sync: torch.cuda.synchronize()
    return mb * reps / (time.perf_counter() - t) / 1e3
print(f"buffer {mb:.1f} MB, torch threads {torch.get_num_threads()}")
print(f"pageable -> pinned  torch.copy_ : {rate(lambda: pin.copy_(src)):6.2f} GB/s")
pn, sn = pin.numpy(), src.numpy()
print(f"pageable -> pinned  np.copyto   : {rate(lambda: np.copyto(pn, sn)):6.2f} GB/s")
for t in (1, 4, 16):
    torch.set_num_threads(t)
    print(f"pageable -> pinned  torch.copy_ ({t:2d} threads): {rate(lambda: pin.copy_(src)):6.2f} GB/s")
print(f"pageable -> pageable np.copyto  : {rate(lambda: np.copyto(np.empty_like(sn), sn)):6.2f} GB/s")
print(f"pinned   -> device  copy_ async : {rate(lambda: dst.copy_(pin, non_blocking=True), sync=True):6.2f} GB/s")
print(f"pageable -> device  copy_       : {rate(lambda: dst.copy_(src), sync=True):6.2f} GB/s")
