#!/usr/bin/env python3
"""Transcribes the per-stage golden tensors of the reference's real-checkpoint test into tests/golden/reference_stage_goldens.json.

Source: /root/reference/src/model/llamatransformer_simulated_test.go:20-1307 (TestSimulatedOnlyFirstLayer / TestSimulatedFull): for the
15-token prompt "What is your name?" through the chat template, the reference pins ~40 intermediate tensors of transformer block 0 --
embedding rows, attention norm, xq / xk / xv, their reshapes, RoPE outputs, repeated and transposed keys / values, scores before and
after the mask, the softmax, attention output before / after wo, h, the block output -- each as a Go literal in "shortened form" (indices
[0, 1, 2, -3, -2, -1] along every axis) with its full size and a tolerance (a multiple of common.THRESHOLD_BF16 / _F32).

Only the VALUES, sizes, tolerances and line numbers are taken (test vectors, like tests/golden/reference_kat.json); the script that
made the file is this one.  tests/test_real_weights.py replays them against the ORACLE's stage dumps on Meta's checkpoint (auto-skipped
without the files, exactly like the reference's own test), so the first machine that has the weights pins every oracle stage, not just
the five output tokens.

    python tests/golden/extract_stage_goldens.py [/root/reference]
"""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
SRC = os.path.join(REF, "src", "model", "llamatransformer_simulated_test.go")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_stage_goldens.json")


def thresholds():
    txt = open(os.path.join(REF, "src", "common", "utils.go")).read()
    th = {}
    for m in re.finditer(r"(THRESHOLD_\w+)\s*=\s*([0-9.eE+-]+)", txt):
        th[m.group(1)] = float(m.group(2))
    return th


def match_brace(txt, i):
    """index just past the brace block that opens at txt[i] == '{'"""
    depth = 0
    while True:
        c = txt[i]
        if c == "{":
            depth += 1
        elif c == "}":
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1


def parse_literal(block):
    """Go composite literal of nested float32 slices -> nested python lists (the /*...,*/ ellipsis markers dropped)"""
    t = re.sub(r"/\*.*?\*/", "", block, flags=re.S)
    t = re.sub(r"//[^\n]*", "", t)
    t = t.replace("{", "[").replace("}", "]")
    t = re.sub(r"\bnegInf\b|float32\(math\.Inf\(-1\)\)|math\.Inf\(-1\)", '"-inf"', t)
    t = re.sub(r",\s*\]", "]", t)
    t = re.sub(r"\[\s*,", "[", t)
    val = json.loads(t)

    def fix(v):
        if isinstance(v, list):
            return [fix(x) for x in v]
        return float(v)
    return fix(val)


def shape_of(v):
    s = []
    while isinstance(v, list):
        s.append(len(v)); v = v[0]
    return s


def main():
    txt = open(SRC).read()
    th = thresholds()
    lines_before = lambda pos: txt.count("\n", 0, pos) + 1
    sizes = {m.group(1): ([int(x) for x in m.group(2).split(",")], lines_before(m.start()))
             for m in re.finditer(r"\bexpected(\w+?)_?Size\s*:=\s*\[\]int\{([\d,\s]+)\}", txt)}
    # tolerance + "shortened" flag of the comparison each golden feeds: CompareTestTensor[Skippable]([skip,] expectedX, expectedXSize, actual, TOL, shortened)
    tol = {}
    for m in re.finditer(r"CompareTestTensor(?:Skippable)?\(\s*(?:\w+\s*,\s*)?expected(\w+?)\s*,\s*expected\w+Size\s*,\s*([\w.\[\]]+)\s*,\s*([^,]+?)\s*,\s*(true|false)\s*\)", txt):
        expr = m.group(3)
        mm = re.match(r"^(?:(\d+)\s*\*\s*)?common\.(THRESHOLD_\w+)$", expr.strip())
        if mm:
            tol.setdefault(m.group(1), {"tolerance": (int(mm.group(1)) if mm.group(1) else 1) * th[mm.group(2)], "expr": expr.strip(),
                                        "shortened": m.group(4) == "true", "actual": m.group(2), "line": lines_before(m.start())})
    out = {}
    for m in re.finditer(r"\bexpected(\w+?)\s*:=\s*((?:\[\])+)float32\{", txt):
        name = m.group(1)
        if name.endswith("Size"):
            continue
        end = match_brace(txt, m.end() - 1)
        vals = parse_literal(txt[m.end() - 1:end])
        key = name[:-len("Shortened")] if name.endswith("Shortened") else name
        size = sizes.get(name, sizes.get(key))
        t = tol.get(name, tol.get(key))
        if size is None or t is None:
            continue                                        # (a literal that no comparison uses)
        out[key] = {"line": lines_before(m.start()), "size": size[0], "shape_given": shape_of(vals), "values": vals,
                    "tolerance": t["tolerance"], "tolerance_expr": t["expr"], "shortened": t["shortened"], "compared_with": t["actual"], "compare_line": t["line"]}
    doc = {"what": "per-stage golden tensors of transformer block 0 for the reference's 15-token test prompt (shortened form: indices 0,1,2,-3,-2,-1 per axis)",
           "source": "src/model/llamatransformer_simulated_test.go (adalkiran/llama-nuts-and-bolts)", "generator": "tests/golden/extract_stage_goldens.py",
           "thresholds": th, "stages": out}
    json.dump(doc, open(OUT, "w"), indent=0, separators=(",", ":"))
    print("wrote %d stage goldens to %s" % (len(out), OUT))
    for k, v in sorted(out.items(), key=lambda kv: kv[1]["line"]):
        print("  :%-5d %-32s size %-16s given %-12s tol %-8g (%s) vs %s" % (v["line"], k, v["size"], v["shape_given"], v["tolerance"], v["tolerance_expr"], v["compared_with"]))


if __name__ == "__main__":
    main()
