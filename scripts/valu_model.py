"""Static VALU count of the headline kernel's plain double sweep, from the compiler's own assembly (no GPU needed):
    python scripts/valu_model.py [out.json]
compiles csrc/ns2d_fast.hip for gfx950 with the library's flags (-S, device only), finds ns2d_fast_sched<float,128,64,16,0,true,0>,
takes its innermost loops that contain exactly the double sweep's 2 x (R + 1) cells -- 68 DPP adds, no LDS reduction -- and counts
their vector instructions (everything SQ_INSTS_VALU counts: v_* including DPP, v_mov, v_readlane).  bench.py multiplies the count
by the run's sweeps to cross-check the instruction figure it rescales from the committed PMC profile."""
import json, os, re, subprocess, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from beacon_amd import build as B

KERNEL = "ns2d_fast_schedIfLi128ELi64ELi16ELi0ELb1ELi0EE"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    asm = "/tmp/valu_model_ns2d_fast.s"
    subprocess.check_call([B.hipcc()] + B.FLAGS + B.FILE_FLAGS.get("ns2d_fast.hip", []) +
                          ["-I", B.INC, "--cuda-device-only", "-S", os.path.join(B.CSRC, "ns2d_fast.hip"), "-o", asm])
    s = open(asm).read()
    m = re.search(r'^(_Z\w*%s\w*):[^\n]*\n(.*?)\n\.Lfunc_end' % KERNEL, s, re.S | re.M)
    lines = [l.strip() for l in m.group(2).split("\n")]
    lines = [l for l in lines if l and not l.startswith(";") and not (l.startswith(".") and not l.startswith(".LBB"))]
    labels = {re.match(r'(\.LBB\d+_\d+):', l).group(1): i for i, l in enumerate(lines) if re.match(r'(\.LBB\d+_\d+):', l)}
    loops = []
    for i, l in enumerate(lines):
        mm = re.match(r's_cbranch_\w+ (\.LBB\d+_\d+)|s_branch (\.LBB\d+_\d+)', l)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in labels and labels[t] < i:
                body = lines[labels[t]:i + 1]
                if any(re.match(r's_cbranch|s_branch', b) for b in body[:-1]):
                    continue                       # not innermost
                valu = sum(b.startswith("v_") for b in body)
                dpp = sum("dpp" in b for b in body)
                ds = sum(b.startswith("ds_") for b in body)
                loops.append({"label": t, "instructions": len(body) - 1, "valu": valu, "dpp": dpp, "lds": ds,
                              "s_nop": sum(b.startswith("s_nop") for b in body)})
    pairs = [lp for lp in loops if lp["dpp"] == 68 and lp["lds"] <= 8]      # 2 x (16 + 1) cells x 2 DPP adds, 4 stores + 4 loads
    res = {"kernel": "ns2d_fast_sched<float,128,64,16,0,true,0>", "hipcc": subprocess.run([B.hipcc(), "--version"], capture_output=True, text=True).stdout.split("\n")[0],
           "plain_double_sweep_loops": pairs, "valu_per_wave_per_plain_double_sweep": max(lp["valu"] for lp in pairs) if pairs else None,
           "waves_per_replica": 8, "cells": 128 * 64,
           "valu_per_replica_sweep_plain": (max(lp["valu"] for lp in pairs) * 8 / 2.0) if pairs else None,
           "source_signature": B.signature()[:16]}
    print(json.dumps(res, indent=1))
    if out:
        with open(out, "w") as fh:
            json.dump(res, fh, indent=1)


if __name__ == "__main__":
    main()
