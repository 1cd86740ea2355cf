import time, numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from oracle import oracle as orc
L = orc.lib()
rng = np.random.default_rng(0)
N, K = 14336, 4096
w = orc.f32_to_bf16((rng.standard_normal((N, K)) * 0.02).astype(np.float32)); x = orc.f32_to_bf16(rng.standard_normal((1, K)).astype(np.float32))
y = np.zeros((1, N), dtype=np.uint16)
for nt in (8, 16, 32, 64, 128, 256):
    L.orc_linear_bf16(orc._p(x), orc._p(w), orc._p(y), 1, N, K, nt)
    t = time.time()
    for _ in range(5): L.orc_linear_bf16(orc._p(x), orc._p(w), orc._p(y), 1, N, K, nt)
    dt = (time.time() - t) / 5
    print("threads %3d: %.2f ms per [14336x4096] GEMV -> %.1f GMAC/s" % (nt, dt * 1e3, N * K / dt / 1e9), flush=True)
