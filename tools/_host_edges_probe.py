import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "flu-plugins-oss_amd")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import pyflutl as F
z = np.load(os.path.join(ROOT, "bench_data", "outline_4k.npz")) if os.path.exists(os.path.join(ROOT, "bench_data", "outline_4k.npz")) else None
if z is None:
    print("no bench outline"); sys.exit(0)
print(z.files)
edges = np.ascontiguousarray(z["edges0"], np.int32); box = tuple(int(v) for v in z["meta"][0][1:5])
w, h = box[2] - box[0], box[3] - box[1]
stream = torch.cuda.current_stream().cuda_stream
dev = torch.from_numpy(edges.view(np.uint8).reshape(-1).copy()).cuda()
host = torch.from_numpy(edges.view(np.uint8).reshape(-1).copy()).pin_memory()
scratch = torch.empty(F.scan_scratch_bytes(len(edges), w, h) + 256, dtype=torch.uint8, device="cuda")
dst = torch.empty(h * (w + 4), dtype=torch.uint8, device="cuda")
ref = None
for name, ed in (("device edges", dev), ("pinned host edges", host), ("device edges", dev), ("pinned host edges", host)):
    for _ in range(3):
        F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, 1)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        F.launch_scan(stream, ed, len(edges), box, 0, scratch, dst, w + 4, 1)
    b.record(); torch.cuda.synchronize()
    out = dst.cpu().numpy().copy()
    if ref is None: ref = out
    print(f"{name}: {a.elapsed_time(b) * 1000 / 50:.1f} us per outline, same coverage: {np.array_equal(out, ref)}")
