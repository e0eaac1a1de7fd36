#!/usr/bin/env python3
"""profiles/pmc_<config>.json from a pmc_summary.txt (tools/collect_profiles.sh): the per-launch counters of the tracker
kernel that bench.py puts beside its live timings.   Usage: tools/pmc_to_json.py <pmc_summary.txt> <config> <source text>"""
import json, re, sys

summary, cfg, source = sys.argv[1], sys.argv[2], sys.argv[3]
blocks, cur = {}, None
for line in open(summary):
    if not line.startswith(" "):
        cur = line.strip()
        blocks[cur] = {}
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean=(\S+)", line)
        if m:
            blocks[cur][m.group(1)] = float(m.group(3))
# the tracker launches of the timed region: the largest grid (joint launches of two segment pairs in the default
# pipeline; the single-pair launches of bench.py's "alone" section and of the first step have half the workgroups)
counts = {}
cur = None
for line in open(summary):
    if not line.startswith(" "):
        cur = line.strip()
    else:
        m = re.match(r"\s+(\S+)\s+n=(\d+)", line)
        if m:
            counts[cur] = max(counts.get(cur, 0), int(m.group(2)))
cands = [k for k in blocks if k.startswith("k_lk_fast") and "true" in k and "workgroups" in k]
lk_name = max(cands, key=lambda k: int(re.search(r"\[(\d+) workgroups", k).group(1))) if cands else next(k for k in blocks if k.startswith("k_lk_fast") and "true" in k)
lk = blocks[lk_name]
kib = 1024.0
out = {
    "source": source,
    "kernel": lk_name,
    "launches_counted": counts.get(lk_name),
    "frame_pairs_per_launch": 2 if len(cands) > 1 else 1,
    # FETCH_SIZE / WRITE_SIZE are reported in KiB per dispatch.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE shows half the
    # bytes of WIDE (16 B/lane) coalesced reads; this kernel gathers single dwords, an access width the guide calls
    # uncalibrated -- the value is given as counted, and doubled as the upper bound
    "lk_fb_fetch_bytes_per_launch": lk.get("FETCH_SIZE", 0) * kib,
    "lk_fb_write_bytes_per_launch": lk.get("WRITE_SIZE", 0) * kib,
    "lk_fb_bytes_per_launch": (lk.get("FETCH_SIZE", 0) + lk.get("WRITE_SIZE", 0)) * kib,
    "lk_fb_bytes_per_launch_fetch_doubled": (2 * lk.get("FETCH_SIZE", 0) + lk.get("WRITE_SIZE", 0)) * kib,
    "lk_fb_valu_insts_per_launch": lk.get("SQ_INSTS_VALU"),
    "lk_fb_salu_insts_per_launch": lk.get("SQ_INSTS_SALU"),
    "lk_fb_lds_insts_per_launch": lk.get("SQ_INSTS_LDS"),
    "lk_fb_waves_per_launch": lk.get("SQ_WAVES"),
    "lk_fb_lds_bank_conflict_ratio": (lk["SQ_LDS_BANK_CONFLICT"] / lk["SQ_LDS_IDX_ACTIVE"]) if lk.get("SQ_LDS_IDX_ACTIVE") else None,
    # share of a wave's lifetime: issuing VALU, stalled at issue, parked at a waitcnt / barrier (quad-cycle counters)
    "lk_fb_wave_time_shares": {k: lk[c] / lk["SQ_WAVE_CYCLES"] for k, c in (("valu_active", "SQ_ACTIVE_INST_VALU"),
                               ("issue_stall", "SQ_WAIT_INST_ANY"), ("parked", "SQ_WAIT_ANY")) if lk.get(c) and lk.get("SQ_WAVE_CYCLES")},
}
if lk.get("GRBM_GUI_ACTIVE") and lk.get("SQ_ACTIVE_INST_VALU"):
    # gfx9 VALUBusy: quad-cycles of VALU issue summed over the SIMDs x 4 / (1024 SIMDs x GPU-active cycles); GRBM_GUI_ACTIVE
    # is the sum over the 8 XCDs
    out["lk_fb_valu_busy_pct"] = 100.0 * lk["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * lk["GRBM_GUI_ACTIVE"] / 8.0)
# the kernels that run beside the tracker launch, per launch: with the tracker's they make up what the SIMDs issue per
# period of the pipeline (one joint tracker launch, one corner kernel, two pyramids, one min-distance chain per two frames)
def insts(prefix):
    b = next((v for k, v in blocks.items() if k.startswith(prefix) and "workgroups" not in k), None)
    return b.get("SQ_INSTS_VALU") if b else None
out["beside_valu_insts_per_launch"] = {"corner_kernel": insts("k_eig_strip") or insts("k_eig_nms"), "k_pyramid_ahead": insts("k_pyramid<3, 64, 64>"),
                                       "min_distance_chain": sum(insts(k) or 0 for k in ("k_key_hist", "k_key_select", "k_cell_count", "k_scan",
                                                                 "k_cell_fill", "k_suppress", "k_gather_accepted", "k_seg_order", "k_seg_init",
                                                                 "k_tail_gather", "k_tail_scatter", "k_tail_rank", "k_tail_order"))}
mw = re.search(r"k_lk_fast(?:88)?<(\d+), (\d+), true>", lk_name)
win = (mw.group(1), mw.group(2)) if mw else ("21", "21")
res = json.load(open("profiles/r04_lk_resources.json"))
rk = "k_lk_fast<%s,%s,true>" % win
out.update(lk_fb_vgprs=res[rk]["vgprs"], lk_fb_waves_per_simd=res[rk]["waves_per_simd"], lk_fb_sgpr_spills=res[rk]["sgpr_spills"],
           lk_fb_scratch_bytes=res[rk].get("scratch_bytes"))
# Mix-weighted VALU issue cost: static instruction classes of the kernel's hot straight-line blocks (tools/isa_mix.py ->
# profiles/r04_isa_mix_lk*.json, r03_isa_mix_strip10 / pyr64 for the kernels that did not change) x the measured issue time of each class (profiles/valu_class_cost.json).  bench.py turns
# it into the peak the achieved rate is divided by: a fraction above 1 would be an accounting error.
cost = json.load(open("profiles/valu_class_cost.json"))["ns_per_inst_per_simd"]
def mix_ns(path):
    try:
        hot = json.load(open(path))["hot_blocks"]
    except (OSError, KeyError):
        return None
    v = {k: n for k, n in hot.items() if k.startswith("valu_")}
    tot = float(sum(v.values()))
    return dict(ns_per_valu_inst=sum(n * cost[k] for k, n in v.items()) / tot, hot_valu_instructions=int(tot),
                hot_class_share={k: n / tot for k, n in v.items()}, hot_salu_per_valu=hot.get("salu", 0) / tot, source=path)
out["valu_mix"] = {"tracker": mix_ns("profiles/r04_isa_mix_lk%s.json" % win[0]), "corner_kernel": mix_ns("profiles/r03_isa_mix_strip10.json"),
                   "pyramid_one_wave": mix_ns("profiles/r03_isa_mix_pyr64.json"),
                   "class_cost_source": "profiles/valu_class_cost.json"}
out["counters_note"] = "counters are from the profiled run named in `source`, not from the run that prints them"
json.dump(out, open("profiles/pmc_%s.json" % cfg, "w"), indent=1)
print(json.dumps(out, indent=1))
