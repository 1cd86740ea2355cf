#!/usr/bin/env python3
"""Rung (a) of the w1|w3 -> w2 streaming-stage ladder (VERDICT r5 task 1; profiles/r06_ffn_stream.md).  Run on the GPU box.

What would a row-band pipeline have to work with -- measured on today's kernels, before anything is built:
 1. both kernel forms (latency / throughput) of gate|up and down, alone;
 2. the PAIR launched on two streams (lnb_profile_ffn_pair), the down kernel `delay` microseconds behind the gate|up kernel:
      delay 0      = plain co-residency (does the dispatcher co-schedule one workgroup of each per CU at all?  r3: no -- measured with the
                     runtime's 4 hardware queues and the 91-124 KB latency forms);
      delay 15..35 = the timeline of a band pipeline whose first row band completes that long after the launch, WITHOUT any dependency stall
                     (w2 reads stale activations: only the time means anything) -- an optimistic bound on what the real thing could reach;
 3. the gate|up kernel in band order (LNB_RW_W13=28: two half-height blocks per workgroup on ONE chain wave).
Every number: microseconds per pair, best of three runs of 48 pairs cycling through 6 layers (weights from HBM)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "llama-nuts-and-bolts_amd")]

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import lnb
    cfg = dict(lnb.LLAMA_8B, n_layers=6)
    m = lnb.LlamaTransformer(**cfg).fill_synthetic(1234).finalize()
    c = lnb.InferenceContext(m, 320)
    _, t = c.Forward(lnb.synth_tokens(99, 128, cfg["vocab_size"]), 0, want_logits=False)
    c.decode_greedy(t, 128, 8)
    sched = os.environ.get("FFN_SCHED", "latency")
    c.set_schedule(sched)
    res = {"schedule": sched, "hw_queues": lnb.runtime_info(0, probe_queues=True)["hw_queues_measured"]}

    def best(f):
        f()
        return round(1e3 * min(f() for _ in range(3)), 2)
    res["w1|w3 alone"] = best(lambda: c.profile_kernel(3, 200, 48))
    res["w2 alone"] = best(lambda: c.profile_kernel(4, 200, 48))
    res["whole block"] = best(lambda: c.profile_kernel(6, 200, 48))
    pads = [int(x) for x in os.environ.get("FFN_PADS", "0").split(",")]
    for pad in pads:
        for delay in [int(x) for x in os.environ.get("FFN_DELAYS", "0,10,15,20,25,30,35").split(",")]:
            res["pair, w2 %d us behind, pad %d" % (delay, pad)] = best(lambda: c.profile_ffn_pair(200, 48, delay, pad))
    print(json.dumps(res))
    sys.exit(0)

out = {}
runs = [("w2 as ONE quad_perm chain wave + 4 helpers (LNB_RW_W2=16 LNB_W2_QUAD=1: 128-step stages, 83 KB), gate|up as in production", {"FFN_SCHED": "latency", "LNB_RW_W2": "16", "LNB_W2_QUAD": "1", "FFN_DELAYS": "-1,0,10,15,20,25,30"}),
        ("... 256-step stages (107 KB: cannot share a CU with gate|up)", {"FFN_SCHED": "latency", "LNB_RW_W2": "16", "LNB_W2_QUAD": "2", "FFN_DELAYS": "0,25"}),
        ("... 128-step stages, gate|up in band order on one chain wave (LNB_RW_W13=28)", {"FFN_SCHED": "latency", "LNB_RW_W2": "16", "LNB_W2_QUAD": "1", "LNB_RW_W13": "28", "FFN_DELAYS": "0,25,30,35"}),
        ("throughput forms, w2's chain waves at s_setprio 3", {"FFN_SCHED": "throughput", "LNB_W2_PRIO": "1", "FFN_DELAYS": "-1,0,10,20,25,30"}),
        ("throughput forms, no priority, w2 launched first", {"FFN_SCHED": "throughput", "FFN_DELAYS": "-1,-10"}),
        ("latency forms (production, 75 + 104 KB of LDS: cannot co-reside)", {"FFN_SCHED": "latency", "FFN_DELAYS": "0,25"}),
        ("throughput forms (75 + 57 KB: one workgroup of each fits a CU)", {"FFN_SCHED": "throughput", "FFN_PADS": "0,28672"}),
        ("throughput forms, 4 hardware queues (the runtime's default)", {"FFN_SCHED": "throughput", "GPU_MAX_HW_QUEUES": "4", "FFN_DELAYS": "0,25"}),
        ("band order on one chain wave (LNB_RW_W13=28), throughput forms", {"FFN_SCHED": "throughput", "LNB_RW_W13": "28", "FFN_DELAYS": "0,25"})]
for label, env in runs:
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    out[label] = json.loads(line[-1]) if line else {"error": r.stderr[-600:]}
print(json.dumps(out, indent=1))
