"""First contact of a multi-GPU run, exercised on the CPU (no GPU, no RCCL): the control star of `bench.py --gpus N` (pipeline.TcpGroup), the
per-rank preflight record, and the rule that ANY rank's problem ends ALL ranks with one message inside the preflight timeout instead of a
hang in the first exchange.  The library is replaced by a stand-in with the four calls the preflight makes."""
import json
import os
import socket
import subprocess
import sys
import textwrap
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "llama-nuts-and-bolts_amd")

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import pipeline
    rank, world, port, scenario = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]

    class FakeLnb:                                      # the calls pipeline.preflight makes (lnb.py: device_count, device_info, can_access_peer, rccl_selftest)
        @staticmethod
        def device_count(): return world
        @staticmethod
        def device_info(d): return {"name": "fake", "arch": "gfx950", "hbm_bytes": 1 << 38, "n_cus": 256}
        @staticmethod
        def can_access_peer(a, b): return True
        @staticmethod
        def pci_bus_id(d): return "0000:{:02x}:00.0".format(0xc1 + (rank if scenario == "one_visible_gpu_per_rank" else d))
        @staticmethod
        def rccl_selftest(d, n):
            if scenario == "selftest_fails_on_rank_1" and rank == 1:
                raise RuntimeError("ncclCommInitRank failed: unhandled system error")

    if scenario == "rank_1_never_comes" and rank == 1:
        sys.exit(0)
    t0 = time.time()
    try:
        grp = pipeline.TcpGroup(rank, world, "127.0.0.1", port, timeout=float(os.environ["LNB_PREFLIGHT_TIMEOUT"]))
    except Exception as e:
        print(json.dumps({"rank": rank, "rendezvous_failed": type(e).__name__, "after_s": round(time.time() - t0, 1)})); sys.exit(4)
    local = 0 if scenario in ("two_ranks_on_one_gpu", "one_visible_gpu_per_rank") else rank
    recs, bad = pipeline.preflight(FakeLnb, grp, rank, world, local)
    print(json.dumps({"rank": rank, "records": recs, "bad": bad}))
    sys.stdout.flush()
    if bad:
        pipeline.abort_all(rank, "in the preflight", bad)
    grp.set_timeout(5.0)
    grp.barrier()
    sys.exit(0)
""") % PKG


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(scenario, world=2, timeout_s="3"):
    port = _free_port()
    env = dict(os.environ, LNB_PREFLIGHT_TIMEOUT=timeout_s)
    procs = [subprocess.Popen([sys.executable, "-c", WORKER, str(r), str(world), str(port), scenario], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    t0 = time.time()
    outs = [p.communicate(timeout=60) for p in procs]
    return [p.returncode for p in procs], outs, time.time() - t0


def test_preflight_records_of_every_rank_reach_every_rank():
    rcs, outs, _ = _run("ok", world=3)
    assert rcs == [0, 0, 0], outs
    for r, (so, _) in enumerate(outs):
        d = json.loads(so.strip().splitlines()[-1])
        assert d["bad"] == [] and [x["rank"] for x in d["records"]] == [0, 1, 2]
        assert all(x["rccl_selftest"].startswith("ok") and x["peer_access"] == [True, True, True] and x["n_cus"] == 256 for x in d["records"])


def test_one_ranks_rccl_failure_ends_all_ranks_with_one_message():
    rcs, outs, secs = _run("selftest_fails_on_rank_1")
    assert rcs == [4, 4] and secs < 20, (rcs, outs)
    assert "ABORTED in the preflight" in outs[0][1] and "rank 1: RuntimeError: ncclCommInitRank failed" in outs[0][1]      # rank 0 prints the one message
    assert "ABORTED" not in outs[1][1]
    for so, _ in outs:                                               # ... and both ranks saw the same verdict
        assert json.loads(so.strip().splitlines()[-1])["bad"] == ["rank 1: RuntimeError: ncclCommInitRank failed: unhandled system error"]


def test_ranks_sharing_a_gpu_are_refused_for_the_rccl_transport():
    rcs, outs, _ = _run("two_ranks_on_one_gpu")
    assert rcs == [4, 4] and "ranks share a GPU" in outs[0][1]


def test_a_rank_that_never_shows_up_ends_the_rendezvous_within_the_timeout():
    rcs, outs, secs = _run("rank_1_never_comes", timeout_s="2")
    assert rcs[0] == 4 and secs < 15, (rcs, outs, secs)
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["rendezvous_failed"] and d["after_s"] <= 5


def test_ranks_that_all_say_device_0_are_told_apart_by_their_pci_bus_id():
    """a launcher that gives every rank ONE visible GPU makes each of them "device 0": the preflight compares PCI bus ids (lnb_device_pci_bus_id), so that
    is not taken for two ranks sharing a GPU -- which still is (test above: same index AND same bus id)"""
    rcs, outs, _ = _run("one_visible_gpu_per_rank", world=2)
    assert rcs == [0, 0], outs
    d = json.loads(outs[0][0].strip().splitlines()[-1])
    assert d["bad"] == [] and [x["device"] for x in d["records"]] == [0, 0] and len({x["pci"] for x in d["records"]}) == 2

