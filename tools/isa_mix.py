#!/usr/bin/env python3
"""Static instruction mix of a gfx950 kernel, from the compiler's own assembly (`hipcc --cuda-device-only -S`).

    python tools/isa_mix.py iceberg_tracking_code_amd/csrc/k_lk_fast.hip 'k_lk_fastILi21ELi21ELb1'  [--blocks] [--json out]

Counts the instructions of the kernel by issue class (the classes of tools/ubench/valu_rate.hip, whose measured issue
costs bench.py weights them with) for the whole kernel and -- with --blocks -- per basic block, marking blocks that
lie inside a loop (a backward branch spans them) with their nesting depth.  No GPU needed.
"""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "--cuda-device-only", "-S"]

# issue classes (tools/ubench/valu_rate.hip measures one representative of each)
FAST = ("v_add_", "v_sub_", "v_subrev_", "v_lshl", "v_lshr", "v_ashr", "v_and_", "v_or_", "v_xor_", "v_mov_", "v_fma_f32", "v_mul_f32",
        "v_add3", "v_max_", "v_min_", "v_max3", "v_min3", "v_med3", "v_bfe", "v_bfi", "v_not_", "v_lshl_add", "v_add_lshl", "v_and_or",
        "v_or3", "v_lshl_or", "v_mac_f32", "v_fmac_f32", "v_cmp_", "v_cmpx_", "v_accvgpr")


def classify(op):
    if op.startswith("v_"):
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu_lane"
        if "_f64" in op or op.startswith(("v_cvt_f64", "v_cvt_i32_f64", "v_cvt_u32_f64")):
            return "valu_f64"
        if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
            return "valu_trans"
        if op.startswith("v_cndmask"):
            return "valu_cndmask"
        if op.startswith(("v_dot", "v_perm", "v_alignb", "v_mad_", "v_mul_i32", "v_mul_u32", "v_mul_lo", "v_mul_hi", "v_pk_", "v_cvt", "v_sad",
                          "v_mbcnt", "v_bcnt", "v_ffb", "v_rndne", "v_floor", "v_ceil", "v_trunc", "v_fract", "v_ldexp", "v_frexp", "v_div_",
                          "v_mqsad", "v_msad", "v_lerp", "v_cubeid")):
            return "valu_slow"
        if op.startswith(FAST):
            return "valu_fast"
        return "valu_other"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_endpgm", "s_setprio", "s_set")):
            return "s_ctrl"
        if op.startswith(("s_cbranch", "s_branch")):
            return "s_branch"
        if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache", "s_memtime", "s_memreal")):
            return "smem"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return "other"


def assemble(src):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [src, "-o", out], stderr=subprocess.DEVNULL, cwd="/tmp")
    return out


def kernel_body(asm, pattern):
    names = re.findall(r"^(_Z\w+):", asm, flags=re.M)
    hits = [n for n in names if pattern in n]
    if not hits:
        raise SystemExit("no kernel matches %r; have: %s" % (pattern, [n[:60] for n in names]))
    name = hits[0]
    start = asm.index("\n" + name + ":")
    end = asm.index(".Lfunc_end", start)
    return name, asm[start:end]


def blocks_of(body):
    """[(label, [ops...], [branch targets])] in program order."""
    out, cur = [], ["entry", [], []]
    for line in body.split("\n")[2:]:
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            out.append(tuple(cur))
            cur = [m.group(1), [], []]
            continue
        t = line.strip()
        if not line.startswith("\t") or not t or t[0] in ".;":
            continue
        op = t.split()[0]
        cur[1].append(op)
        if op.startswith(("s_cbranch", "s_branch")):
            cur[2].append(t.split()[1])
    out.append(tuple(cur))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source", help=".hip file (compiled here) or a .s file")
    ap.add_argument("kernel", help="substring of the mangled kernel name")
    ap.add_argument("--blocks", action="store_true")
    ap.add_argument("--json")
    ap.add_argument("--hot-min", type=int, default=40,
                    help="basic blocks inside loops with at least this many instructions count as the hot straight-line code "
                         "(the reflecting border loaders and other cold paths are many small blocks)")
    args = ap.parse_args()
    path = args.source if args.source.endswith(".s") else assemble(os.path.abspath(args.source))
    name, body = kernel_body(open(path).read(), args.kernel)
    bl = blocks_of(body)
    index = {b[0]: i for i, b in enumerate(bl)}
    depth = [0] * len(bl)
    for i, (_, _, targets) in enumerate(bl):
        for t in targets:
            j = index.get(t)
            if j is not None and j <= i:      # backward branch: blocks j..i form a loop
                for k in range(j, i + 1):
                    depth[k] += 1
    total = collections.Counter()
    in_loop = collections.Counter()
    per_block = []
    for (label, ops, _), d in zip(bl, depth):
        c = collections.Counter(classify(o) for o in ops)
        total.update(c)
        if d:
            in_loop.update(c)
        per_block.append(dict(label=label, loop_depth=d, n=len(ops), classes=dict(c)))
    hot = collections.Counter()
    for b in per_block:
        if b["loop_depth"] and b["n"] >= args.hot_min:
            hot.update(b["classes"])
    res = dict(kernel=name, total=dict(total), inside_loops=dict(in_loop), hot_blocks=dict(hot), hot_min=args.hot_min,
               instructions=sum(total.values()))
    print(json.dumps(res, indent=1))
    if args.blocks:
        for b in per_block:
            if b["n"]:
                print("%-12s depth %d  n %5d  %s" % (b["label"], b["loop_depth"], b["n"], b["classes"]))
    if args.json:
        res["blocks"] = per_block
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
