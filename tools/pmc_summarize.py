#!/usr/bin/env python3
"""Aggregate the per-dispatch counter CSVs written by tools/pmc_profile.sh by kernel name.
usage: python tools/pmc_summarize.py gpurun_out/pmc_<tag> [out.md]"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(conv_h3_kernel|conv_h2_kernel|conv_bneck_kernel|conv_stem2_kernel|conv_pw_kernel|cls_mega_kernel|sppf3_kernel|greedy_nmm_kernel|zero_i32_kernel|conv_t2d_kernel|conv_halop_kernel|conv_halo_kernel|conv_dmap_kernel|conv_dmh_kernel|conv_ws_kernel|conv_dma_kernel|conv_igemm_kernel|stem_kernel|maxpool5_kernel|decode_kernel|"
                  r"nms_sort_greedy_kernel|nms_prefilter_kernel|cls_head_kernel)", name)
    if not m:
        return name[:40]
    k = m.group(1)
    if k in ("conv_h3_kernel", "conv_bneck_kernel", "conv_stem2_kernel"):
        nums = re.findall(r"Li(\d+)E", name) or re.findall(r"[<,] ?(\d+)(?=[,>])", name)
        return f"{k.replace('_kernel', '')}<f16,{','.join(nums)}>"
    d = re.search(r"<(_Float16|float|miyolo::fp8_t)((?:, \w+)*)>", name)          # demangled form (rocprofv3 prints either)
    if d:
        dt = {"_Float16": "f16", "float": "f32", "miyolo::fp8_t": "f8"}[d.group(1)]
        nums = [x for x in re.findall(r", (\w+)", d.group(2)) if x.isdigit()]
        return f"{k.replace('_kernel', '')}<{dt},{','.join(nums)}>"
    t = re.search(r"I(DF16_|f|NS_5fp8_tE)((?:Li\d+E)*)", name)
    if t:
        dt = "f16" if t.group(1) == "DF16_" else "f8" if "fp8" in t.group(1) else "f32"
        nums = re.findall(r"Li(\d+)E", t.group(2))
        return f"{k.replace('_kernel', '')}<{dt},{','.join(nums)}>"
    return k


def main():
    root = sys.argv[1]
    data = defaultdict(lambda: defaultdict(float))
    counts = defaultdict(int)
    dur = defaultdict(float)
    for g in sorted(os.listdir(root)):
        files = sorted(glob.glob(os.path.join(root, g, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]   # newest run only
        for f in files:
            seen = set()
            for row in csv.DictReader(open(f)):
                k = short(row["Kernel_Name"])
                data[k][row["Counter_Name"]] += float(row["Counter_Value"])
                key = (g, row["Dispatch_Id"])
                if g == "sq1" and key not in seen:
                    seen.add(key)
                    counts[k] += 1
        for f in sorted(glob.glob(os.path.join(root, g, "*", "*kernel_trace.csv")), key=os.path.getmtime)[-1:]:
            if g != "sq1":
                continue
            for row in csv.DictReader(open(f)):
                dur[short(row["Kernel_Name"])] += (float(row["End_Timestamp"]) - float(row["Start_Timestamp"])) * 1e-6
    lines = ["| kernel | launches | ms | MFMA util % | wait_any % | wait_inst % | LDS conflict % | L2 hit % | "
             "fetch GB (x2) | write GB | HBM rd GB/s | TA rd req/launch |", "|" + "---|" * 12]
    for k in sorted(data, key=lambda x: -dur.get(x, 0)):
        d = data[k]
        n = max(counts[k], 1)
        ms = dur.get(k, 0.0)
        busy = d.get("SQ_BUSY_CYCLES", 0)
        wc = d.get("SQ_WAVE_CYCLES", 0)
        # SQ_VALU_MFMA_BUSY_CYCLES sums matrix-pipe busy cycles over the 1024 SIMDs; GRBM_GUI_ACTIVE sums shader
        # clocks over the 8 XCDs (separate pass): utilisation = busy / (1024 * GRBM/8)
        gui = d.get("GRBM_GUI_ACTIVE", 0) / 8.0
        mfma = 100 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024.0 * gui) if gui else 0
        wa = 100 * d.get("SQ_WAIT_ANY", 0) / wc if wc else 0
        wi = 100 * d.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0
        ldsc = 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / d.get("SQ_LDS_IDX_ACTIVE", 1) if d.get("SQ_LDS_IDX_ACTIVE") else 0
        hit = 100 * d.get("TCC_HIT_sum", 0) / (d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0)) if d.get("TCC_HIT_sum") else 0
        fetch = 2 * d.get("FETCH_SIZE", 0) * 1024 / 1e9      # KB units; x2: gfx950 counts 128-B requests as 64 B
        write = d.get("WRITE_SIZE", 0) * 1024 / 1e9
        bw = fetch / (ms * 1e-3) if ms else 0
        lines.append(f"| {k} | {n} | {ms:.2f} | {mfma:.1f} | {wa:.1f} | {wi:.1f} | {ldsc:.1f} | {hit:.1f} | {fetch:.2f} | "
                     f"{write:.2f} | {bw:.0f} | {d.get('TCP_TCC_READ_REQ_sum', 0) / n:.0f} |")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()
