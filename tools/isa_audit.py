#!/usr/bin/env python3
"""Static audit of the hand-counted register rings in gemv_chain_kernel, rowcast_kernel, attn_exact_kernel and mfma_stream_kernel
(llama-nuts-and-bolts_amd/csrc/lnb_kernels.hip, lnb_batch_kernels.h).

The helper waves issue `global_load_dwordx4 ... nt` from inline asm (invisible to hipcc's s_waitcnt bookkeeping) and
retire them with a hand-written counted `s_waitcnt vmcnt(N)`.  hipcc is free to copy / reuse VGPRs it believes are
already written, so between an asm load and the asm wait that retires it NO compiler instruction may touch the
destination registers (cdna_hip_programming.md section 5.7 item 1).  The asm statements tag themselves
("; RING_LOAD", "; RING_RETIRE v[a:b] ...", "; RING_RETIRE_ALL"); this script builds the CFG of every
gemv_chain_kernel instantiation from the -save-temps .s and runs a forward may-analysis (in-flight register set,
union at joins, to a fixpoint): any non-asm instruction that mentions an in-flight VGPR is a violation.
Also requires zero scratch / spills and no compiler v_accvgpr traffic.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "llama-nuts-and-bolts_amd", "csrc", "lnb_kernels.hip")


def compile_to_asm(workdir):
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.dirname(SRC), "-c", SRC, "-save-temps=obj", "-o", os.path.join(workdir, "k.o")]
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for f in os.listdir(workdir):
        if f.endswith("gfx950.s"):
            return os.path.join(workdir, f)
    raise RuntimeError("no gfx950 .s produced")


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


class Block:
    def __init__(self, label):
        self.label, self.ins, self.succ = label, [], []   # ins: (lineno, text, is_asm)


def build_cfg(lines, execz_both=False):
    """basic blocks split at every label AND after every branch instruction"""
    blocks, order = {}, []
    counter = [0]

    def new_block(label=None):
        if label is None:
            counter[0] += 1
            label = "<anon%d>" % counter[0]
        b = Block(label); blocks[label] = b; order.append(b)
        return b

    cur = new_block("<entry>")
    in_asm = False
    for no, ln in lines:
        t = ln.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = new_block(m.group(1))
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True; continue
        if t.startswith(";;#ASMEND"):
            in_asm = False; continue
        if in_asm and t.startswith(";") and "RING_" in t:      # a marker-only asm statement (wait_slot's per-register retires)
            cur.ins.append((no, "s_nop 0 " + t, True)); continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        cur.ins.append((no, t, in_asm))
        if not in_asm and (t.startswith("s_cbranch") or t.startswith("s_branch") or t.startswith("s_endpgm")):
            cur = new_block()
    # successors: EXEC is never zero in a running wave of this kernel (the helper waves run with all 64 lanes active),
    # so the structurizer's `s_cbranch_execz` skip-edges are never taken and `s_cbranch_execnz` always is
    for i, b in enumerate(order):
        fall = True
        if b.ins:
            t = b.ins[-1][1]
            m = re.match(r"^s_cbranch_execz\s+(\.LBB\d+_\d+)", t)
            m2 = re.match(r"^s_cbranch_execnz\s+(\.LBB\d+_\d+)", t)
            m3 = re.match(r"^s_cbranch_\w+\s+(\.LBB\d+_\d+)", t)
            m4 = re.match(r"^s_branch\s+(\.LBB\d+_\d+)", t)
            if m:
                if execz_both:
                    b.succ.append(m.group(1))
            elif m2:
                b.succ.append(m2.group(1)); fall = False
            elif m3:
                b.succ.append(m3.group(1))
            elif m4:
                b.succ.append(m4.group(1)); fall = False
            elif t.startswith("s_endpgm"):
                fall = False
        if fall and i + 1 < len(order):
            b.succ.append(order[i + 1].label)
    return blocks, order


def transfer(b, state, report):
    st = set(state)
    for no, t, is_asm in b.ins:
        code = t.split(";")[0]
        if is_asm:
            if "RING_LOAD" in t:
                st |= regs_of(code.split()[1])
            elif "RING_RETIRE_ALL" in t:
                st = set()
            elif "RING_RETIRE" in t:
                st -= regs_of(t.split("RING_RETIRE")[1])
            continue
        if st:
            bad = regs_of(code) & st
            if bad and report is not None:
                report.append((no, code.strip(), sorted(bad)))
    return st


def audit_function(lines, execz_both=False):
    blocks, order = build_cfg(lines, execz_both)
    inn = {b.label: set() for b in order}
    work = [order[0].label]
    seen_in = {order[0].label: set()}
    while work:
        lab = work.pop()
        out = transfer(blocks[lab], inn[lab], None)
        for s in blocks[lab].succ:
            if s not in blocks:
                continue
            if s not in seen_in or not out <= inn[s]:
                inn[s] = inn[s] | out
                seen_in[s] = True
                work.append(s)
    viol = []
    for b in order:
        if b.label in seen_in:
            transfer(b, inn[b.label], viol)
        elif any(is_asm and "RING_" in t for _, t, is_asm in b.ins):
            # an analysis that never reaches the ring code would pass vacuously
            viol.append((b.ins[0][0], "UNREACHABLE block with ring code: " + b.label, []))
    return viol


def main(asm_path=None):
    with tempfile.TemporaryDirectory() as wd:
        asm = asm_path if asm_path else compile_to_asm(wd)
        txt = open(asm).read().split("\n")
    funcs, cur = {}, None
    for i, ln in enumerate(txt):
        m = re.match(r"^(_Z\d\d(?:gemv_chain|gemv_quad|attn_exact|attn_gqa|attn_long_scores|attn_mfma3|rowcast|rowcast_lds|mfma_stream|mfma_pair|gemm_stream)_kernel\S*):", ln)
        if m:
            cur = []; funcs[m.group(1)] = cur
            continue
        if cur is not None:
            cur.append((i + 1, ln))
            if "s_endpgm" in ln and ln.strip().startswith("s_endpgm") and False:
                pass
            if ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
                cur = None
    total = 0
    for name, lines in funcs.items():
        # gemv_chain_kernel: the ring lives in helper waves that always run with a full EXEC mask, so the structurizer's
        # execz skip-edges are dead there; the other kernels are analysed with both edges
        v = audit_function(lines, execz_both=not (name.startswith("_Z17gemv_chain") or name.startswith("_Z16gemv_quad")))
        reach = sum(1 for _, l in lines if "RING_RETIRE" in l)
        body = "\n".join(l for _, l in lines)
        n_loads = body.count("RING_LOAD")
        accv = sum(1 for _, l in lines if "v_accvgpr" in l)
        print("%-92s ring loads %3d  violations %d  v_accvgpr %d" % (name[:92], n_loads, len(v), accv))
        for no, t, bad in v[:6]:
            print("    line %d: %s   <- in-flight v%s" % (no, t, bad))
        # (mfma_stream_kernel's accumulators LIVE in AGPRs -- the matrix cores write them there: accvgpr moves are its epilogue, not a spill)
        total += len(v) + (0 if ("mfma_stream" in name or "mfma_pair" in name or "gemm_stream" in name) else accv)
    # scratch memory and VGPR spills: never.  SGPR spills into VGPR lanes (v_writelane, no memory): tolerated for the kernels listed here only --
    # attn_gqa_kernel inlines the f64 exp (two dozen SGPRs of polynomial constants) and saves 18 scalars in one VGPR around it
    # (round 6: attn_one_kernel inlines the same exp beside its in-launch exchange: 30 scalars parked in VGPR lanes; it is the opt-in one-launch form)
    SGPR_SPILL_OK = ("attn_gqa_kernel", "attn_one_kernel")
    spills, cur = [], ""
    for l in txt:
        mname = re.search(r"\.name:\s+(\S+)", l)
        if mname: cur = mname.group(1)
        if re.search(r"\.vgpr_spill_count:\s+[1-9]", l) or re.search(r"\.private_segment_fixed_size:\s+[1-9]", l): spills.append(l)
        elif re.search(r"\.sgpr_spill_count:\s+[1-9]", l) and not any(k in cur for k in SGPR_SPILL_OK): spills.append(l)
    print("TOTAL violations:", total, "in", len(funcs), "kernels; spill/scratch metadata lines:", len(spills))
    return 1 if (total or spills or not funcs) else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else None))
