#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/leak.py <<'PY'
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "llama-nuts-and-bolts_amd")); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, ctypes as C
import lnb
from oracle import oracle as orc
TINY = dict(orc.TINY)
hip = C.CDLL("libamdhip64.so")
def free():
    f, t = C.c_size_t(0), C.c_size_t(0); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value
gm = lnb.LlamaTransformer(**TINY).fill_synthetic(1234).finalize()
prompt = orc.synth_tokens(31, 10, TINY["vocab_size"])
def gen():
    gc = lnb.InferenceContext(gm, 64)
    _, first = gc.Forward(prompt, 0, want_logits=True)
    got, _ = gc.decode_greedy(first, 10, 6)
    gc.set_attention(0, 0)
    gc.decode_greedy(int(got[-1]), 16, 6)
    pipe = lnb.Pipeline(gm, 0, 1, None); pc = lnb.InferenceContext(gm, 64)
    pipe.tick(run=pc, run_rows=10, run_pos=0, run_tokens=np.ascontiguousarray(prompt, dtype=np.int32))
    for i in range(3): pipe.tick(run=pc, run_rows=1, run_pos=10 + i)
    pipe.sync(); pipe.close(); pc.close(); gc.close()
gen(); gen()
f0 = free()
for n in (25, 100, 400):
    for _ in range(n): gen()
    print(n, "generations:", (f0 - free()) / 1048576.0, "MB below the start")
import resource
print("host maxrss MB", resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0)
PY
python /tmp/leak.py
