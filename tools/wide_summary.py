#!/usr/bin/env python3
"""Merge the counter summaries that tools/profile_wide.sh leaves under gpurun_out/wide_<tag>/ into one JSON with
the derived shares per kernel.  usage: tools/wide_summary.py gpurun_out/wide_<tag> > profiles/<name>.json"""
import json
import os
import re
import sys

KERNELS = (("amp", "amp_step_pair_kernel<2> (2^20 envs)", "amp_step_pair_kernel"),
           ("allegro", "allegro_post_kernel<full_state,64> (2^20 envs)", "allegro_post_kernel"))
N = 1 << 20


def main():
    root = sys.argv[1]
    stats = open(os.path.join(root, "kernel_stats_tasks.txt")).read()
    out = {}
    for key, title, match in KERNELS:
        c = {}
        for part in ("sq", "traffic"):
            c.update(json.load(open(os.path.join(root, "%s_%s" % (part, key), "summary.json"))))
        m = re.search(re.escape(match) + r".*?calls\s+(\d+)\s+avg\s+([0-9.]+) us", stats)
        calls, us = int(m.group(1)), float(m.group(2))
        rd, wr = c["TCC_EA0_RDREQ_128B_sum"] * 128.0, c["TCC_EA0_WRREQ_64B_sum"] * 64.0
        wc = c["SQ_WAVE_CYCLES"]
        c["derived"] = {
            "valu_busy_share_of_wave_cycles": round(c["SQ_ACTIVE_INST_VALU"] / wc, 4),
            "wait_on_memory_or_barrier_share (SQ_WAIT_ANY)": round(c["SQ_WAIT_ANY"] / wc, 4),
            "issue_stall_share (SQ_WAIT_INST_ANY)": round(c["SQ_WAIT_INST_ANY"] / wc, 4),
            "lds_active_share": round(c["SQ_ACTIVE_INST_LDS"] / wc, 4),
            "lds_conflict_cycles_over_lds_cycles": round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4),
            "valu_instructions_per_wave": round(c["SQ_INSTS_VALU"] / c["SQ_WAVES"], 1),
            "hbm_read_bytes_per_env": round(rd / N, 1),
            "hbm_write_bytes_per_env": round(wr / N, 1),
            "hbm_bytes_per_env": round((rd + wr) / N, 1),
            "kernel_us_same_box (rocprofv3 kernel trace, %d launches)" % calls: us,
            "TBps_on_delivered_bytes": round((rd + wr) / us / 1e6, 3),
        }
        out[title] = c
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
