import torch, time
dev = torch.device("cuda:0")
n = 256 * 2 * 2 * 720 * 1280            # the four flow planes of 256 frame pairs (floats)
a = torch.empty(n, dtype=torch.float32, device=dev)
m = torch.empty(256 * 2 * 720 * 1280, dtype=torch.uint8, device=dev)
src = torch.empty(n, dtype=torch.float32, device=dev)
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(it)]
    for s, e in ev:
        s.record(); fn(); e.record()
    torch.cuda.synchronize()
    ts = sorted(s.elapsed_time(e) for s, e in ev)
    return ts[len(ts) // 2]
ms = t(lambda: (a.fill_(1.5), m.fill_(1)))
gb = (n * 4 + m.numel()) / 1e9
print(f"pure write of the flow_prepare outputs ({gb:.2f} GB): {ms:.3f} ms = {gb / ms:.2f} TB/s")
ms = t(lambda: a.copy_(src))
print(f"copy (read+write {2 * n * 4 / 1e9:.2f} GB): {ms:.3f} ms = {2 * n * 4 / 1e9 / ms:.2f} TB/s")
