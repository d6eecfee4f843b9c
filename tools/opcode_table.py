"""Per-opcode VALU count of the headline kernel's interior chunk (32 unrolled K = 8 steps of pa::pair_kernel<8, false, true>), from the ISA.
Usage:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -c astar-pairwise-aligner_amd/csrc/pa_hip.hip -o /tmp/pa.o -save-temps
        python tools/opcode_table.py pa_hip-hip-amdgcn-amd-amdhsa-gfx950.s > profiles/r05_headline_opcodes.json
Classes (measured, DESIGN.md 2 "Roofline", tools/issue_probe.py): FAST = unbroken VOP2 logic / add without carry, v_lshrrev imm, v_mov,
3-VGPR v_bitop3 / v_and_or (2.3-3 clocks per wave64 instruction per SIMD); SLOW = v_alignbit, v_bfe, v_lshl_or, v_add_co / v_addc, DPP,
v_readlane, anything with an SGPR / VCC source operand (4.1-4.3 clocks)."""
import collections
import json
import re
import sys

SYMBOL = r"_ZN2pa11pair_kernelILi8ELb0ELb1EEEvPKNS_8StripJobEPKiiPj"
SLOW = {"v_alignbit_b32", "v_bfe_u32", "v_lshl_or_b32", "v_add_co_u32_e32", "v_add_co_u32_e64", "v_addc_co_u32_e32", "v_addc_co_u32_e64", "v_mov_b32_dpp",
        "v_readlane_b32", "v_lshlrev_b32_e32", "v_add3_u32", "v_lshl_add_u64"}


def main():
    txt = open(sys.argv[1]).read()
    m = re.search(r"^(" + SYMBOL + r"):", txt, re.M)
    body = txt[m.end():txt.index(".Lfunc_end", m.end())].split("\n")
    blocks, cur = [], []
    for ln in body:
        t = ln.strip()
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = []
        else:
            cur.append(t)
    blocks.append(cur)
    instr = lambda b: [x for x in b if x and re.match(r"^[a-z]", x)]
    # the interior chunk: the largest basic block without per-step predication (no v_cndmask_b32_e32 chain)
    cand = [b for b in blocks if sum(x.startswith("ds_read_b128") for x in b) == 64 and sum(x.startswith("v_cndmask_b32_e32") for x in b) < 32]
    b = max(cand, key=lambda b: len(instr(b)))
    valu = [x for x in instr(b) if x.startswith("v_")]

    def sgpr_source(x):
        parts = [p.strip() for p in (x.split(None, 1)[1] if " " in x else "").split(",")]
        return any(re.match(r"^(s\d+|s\[\d+:\d+\]|vcc|vcc_lo|exec)$", p) for p in parts[1:])

    counts = collections.Counter(x.split()[0] for x in valu)
    slow = sum(1 for x in valu if x.split()[0] in SLOW or sgpr_source(x))
    fast = len(valu) - slow
    steps = 32
    out = {"kernel": "pa::pair_kernel<8, false, true>", "block": "interior chunk (32 unrolled steps, no predication)", "steps": steps,
           "valu_per_step": round(len(valu) / steps, 2), "fast_per_step": round(fast / steps, 2), "slow_per_step": round(slow / steps, 2),
           "per_step_by_opcode": {k: round(v / steps, 2) for k, v in sorted(counts.items(), key=lambda kv: -kv[1])},
           "sgpr_or_vcc_source_per_step": round(sum(1 for x in valu if sgpr_source(x)) / steps, 2),
           "other": {"ds_read_b128": sum(x.startswith("ds_read_b128") for x in b), "s_nop": sum(x.startswith("s_nop") for x in b),
                     "s_waitcnt": sum(x.startswith("s_waitcnt") for x in b)},
           "clocks": {"fast": [2.3, 3.0], "slow": [4.1, 4.3]}}
    f, s = out["fast_per_step"], out["slow_per_step"]
    out["weighted_clocks_per_instruction"] = [round((f * 2.3 + s * 4.1) / (f + s), 3), round((f * 3.0 + s * 4.3) / (f + s), 3)]
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
