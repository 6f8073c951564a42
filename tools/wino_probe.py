"""Which MIOpen solvers exist for the 3x3 fp32 convolutions of ResNet-50 on this GPU, per layout (find mode, logged)?"""
import os, sys, time
os.environ.setdefault("MIOPEN_FIND_MODE", "1")
os.environ.setdefault("MIOPEN_FIND_ENFORCE", "3")      # search, ignore the user db
os.environ["MIOPEN_USER_DB_PATH"] = "/tmp/miopen_probe_db"
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = "/tmp/miopen_probe_db"
os.makedirs("/tmp/miopen_probe_db", exist_ok=True)
import torch
import torch.nn.functional as F
torch.backends.cudnn.benchmark = True
dev = "cuda"
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for (C, H) in ((64, 56), (128, 28), (256, 14), (512, 7)):
    for cl in (False, True):
        x = torch.randn(N, C, H, H, device=dev)
        w = torch.randn(C, C, 3, 3, device=dev)
        if cl:
            x = x.contiguous(memory_format=torch.channels_last); w = w.contiguous(memory_format=torch.channels_last)
        x.requires_grad_(True); w.requires_grad_(True)
        for _ in range(3):
            y = F.conv2d(x, w, padding=1)
            y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        reps = 10
        ev[0].record()
        for _ in range(reps):
            y = F.conv2d(x, w, padding=1)
        ev[1].record()
        g = torch.ones_like(y)
        for _ in range(reps):
            torch.autograd.grad(y, (x, w), g, retain_graph=True)
        ev[2].record()
        torch.cuda.synchronize()
        fl = 2 * N * C * C * 9 * H * H
        tf, tb = ev[0].elapsed_time(ev[1]) / reps, ev[1].elapsed_time(ev[2]) / reps
        print(f"C={C} H={H} {'NHWC' if cl else 'NCHW'}: fwd {tf:.3f} ms ({fl / tf / 1e9:.1f} TF/s direct-equivalent)  bwd(d+w) {tb:.3f} ms ({2 * fl / tb / 1e9:.1f} TF/s)", flush=True)
