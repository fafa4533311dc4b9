"""Turn gpurun_out/<tag>_profile/ (tools/round4_profile.sh) into the files under profiles/: kernel stats csv, the PMC sums as
text, and profiles/traffic.json (per-launch HBM bytes, L2 requests / misses and vector instructions per read for the kernels
bench.py quotes).  usage: python tools/summarise_profile.py <tag> <round-label> [reads-per-launch]"""
import ast, csv, json, os, shutil, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, label = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000_000
src = os.path.join(ROOT, "gpurun_out", tag + "_profile")
dst = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(dst, label + "_kernel_stats.csv"))
shutil.copy(os.path.join(src, "pmc.txt"), os.path.join(dst, label + "_pmc.txt"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, label + "_bench.json"))
pmc = {}
for l in open(os.path.join(src, "pmc.txt")):
    name, d = l.split(" {", 1)
    d = ast.literal_eval("{" + d)
    pmc.setdefault(name.strip(), {}).update({k: v["mean_per_dispatch"] for k, v in d.items()})
stats = {r["Name"].split("(")[0].strip(): float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(os.path.join(src, "kernel_stats.csv")))}
def find(prefix, table):
    for k, v in table.items():
        if prefix in k:
            return v
    return None
def kernel(prefix):
    c = find(prefix, pmc) or {}
    out = {"avg_ms": find(prefix, stats)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        out["hbm_bytes_per_launch"] = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024  # (KB as counted: no gfx950 correction for 64-byte gathers)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        out["l2_requests_per_read"] = (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) / B
        out["l2_misses_per_read"] = c["TCC_MISS_sum"] / B
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
              "SQ_THREAD_CYCLES_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
        if k in c:
            out[k] = c[k]
    return out
k = {n: kernel(p) for n, p in (("k_gapped_rows", "k_gapped_rows<160"), ("k_seed_extend", "k_seed_extend<false, 3, true, false>"),
                               ("k_sort_consensus_32", "k_sort_consensus<32>"), ("k_sort_consensus_64", "k_sort_consensus<64>"),
                               ("k_reg_scatter", "k_reg_scatter"), ("k_seg_scatter", "k_seg_scatter"), ("k_reg_hist", "k_reg_hist"), ("k_seg_hist", "k_seg_hist"),
                               ("k_dust_scan", "k_dust_scan"), ("k_dust_perfect", "k_dust_perfect"), ("k_dust_windows", "k_dust_windows"))}
bench = json.load(open(os.path.join(src, "bench.json")))
gap_ms = bench["stages_ms_last_step"]["gapped"]
stage_kernels = ("k_gapped_rows", "k_reg_scatter", "k_seg_scatter", "k_reg_hist", "k_seg_hist")
tj = {
    "round": label,
    "how": "tools/round4_profile.sh + tools/summarise_profile.py: rocprofv3 --kernel-trace --stats on bench.py --steps 5 --warmup 1, and six rocprofv3 --pmc passes "
           "(one counter group each, never with tracing) on tools/quick_bench.py 10000000 2, i.e. at the bench's own 10 M reads per launch (the region order of "
           "the gapped stage depends on the launch size); values = mean per dispatch.  FETCH_SIZE / WRITE_SIZE are quoted as counted (KB x 1024): the gfx950 "
           "correction of the guide is calibrated for wide coalesced streams, these kernels move 16-64-byte gathers.",
    "reads_per_launch": B,
    "kernels": k,
    "k_seed_extend_bytes_per_launch": k["k_seed_extend"].get("hbm_bytes_per_launch"),
    "k_seed_extend_l2_requests_per_read": k["k_seed_extend"].get("l2_requests_per_read"),
    "k_gapped_stage_bytes_per_launch": sum((k[n].get("hbm_bytes_per_launch") or 0) for n in stage_kernels),
    "k_gapped_rows_l2_misses_per_read": k["k_gapped_rows"].get("l2_misses_per_read"),
    "k_gapped_rows_l2_requests_per_read": k["k_gapped_rows"].get("l2_requests_per_read"),
    "k_gapped_rows_valu_instructions_per_read": (k["k_gapped_rows"].get("SQ_INSTS_VALU") or 0) / B,
    "k_gapped_rows_share_of_stage": (k["k_gapped_rows"]["avg_ms"] or 0) / gap_ms if gap_ms else None,
    "k_sort_consensus_bytes_per_launch": (k["k_sort_consensus_32"].get("hbm_bytes_per_launch") or 0) + (k["k_sort_consensus_64"].get("hbm_bytes_per_launch") or 0),
}
json.dump(tj, open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps({n: {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if a in ("avg_ms", "hbm_bytes_per_launch", "l2_requests_per_read", "l2_misses_per_read")} for n, v in k.items()}, indent=1))
