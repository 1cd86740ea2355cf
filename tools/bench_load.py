#!/usr/bin/env python3
"""Weight-ingestion throughput: write a PyTorch zip checkpoint of Llama-3.1-8B-shaped blocks with torch.save, then time
lnb_checkpoint_open (mmap + zip directory + pickle VM) and lnb_model_load_checkpoint (page cache -> HBM + re-tiling).
usage: python tools/bench_load.py [n_layers=4] [dir=/tmp]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "llama-nuts-and-bolts_amd"))
import torch  # noqa: E402
import lnb  # noqa: E402

n_layers = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = sys.argv[2] if len(sys.argv) > 2 else "/tmp"
cfg = dict(lnb.LLAMA_8B); cfg["n_layers"] = n_layers
m = lnb.LlamaTransformer(device=0, **cfg)
infos = m.tensor_infos()
g = torch.Generator().manual_seed(1)
t0 = time.time()
sd = {name: (torch.randn(*shape, generator=g, dtype=torch.float32) * 0.02).to(torch.bfloat16) for name, shape in infos}
path = os.path.join(d, "bench_load.pth")
torch.save(sd, path)
size = os.path.getsize(path)
del sd
print("wrote %s: %.2f GB, %d tensors in %.1f s" % (path, size / 1e9, len(infos), time.time() - t0))
for rep in range(2):                                   # second pass: file fully in the page cache
    t0 = time.perf_counter()
    ck = lnb.Checkpoint(path)
    t1 = time.perf_counter()
    m.load_checkpoint(ck)
    t2 = time.perf_counter()
    ck.close()
    print("pass %d: open+unpickle %.1f ms (%d tensors); bind (host -> HBM + re-tile) %.2f s = %.2f GB/s" %
          (rep, 1e3 * (t1 - t0), len(infos), t2 - t1, size / 1e9 / (t2 - t1)))
m.close()
os.remove(path)
