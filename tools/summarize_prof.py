#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd (sqlite) outputs: per-kernel duration stats and per-kernel PMC means.

    python tools/summarize_prof.py gpurun_out/prof_<tag>  > profiles/rNN_<tag>_summary.txt
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name if len(name) <= 84 else name[:81] + "..."


def kernel_stats(db):
    cur = db.cursor()
    rows = cur.execute(
        "select s.kernel_name, count(*), sum(d.end - d.start), avg(d.end - d.start), min(d.end - d.start), max(d.end - d.start), "
        "max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size), max(d.grid_size_x*d.grid_size_y*d.grid_size_z), max(d.workgroup_size_x) "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = []
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, grid, wg in rows[:16]:
        out.append("  {:<84s} calls {:>4d}  avg_us {:>10.2f}  min_us {:>10.2f}  max_us {:>10.2f}  pct {:>5.1f}  vgpr {:>3} agpr {:>3} sgpr {:>3} lds {:>6} grid_threads {:>8} wg {}".format(
            short(name), n, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds, grid, wg))
    return out


def counters(db):
    cur = db.cursor()
    try:
        rows = cur.execute(
            "select s.kernel_name, p.name, avg(e.value), count(*) from rocpd_pmc_event e "
            "join rocpd_info_pmc p on e.pmc_id = p.id "
            "join rocpd_kernel_dispatch d on e.event_id = d.event_id "
            "join rocpd_info_kernel_symbol s on d.kernel_id = s.id group by s.kernel_name, p.name").fetchall()
    except sqlite3.Error as e:
        return ["  (no counters: {})".format(e)]
    acc = defaultdict(dict)
    for k, c, v, n in rows:
        acc[short(k)][c] = (v, n)
    out = []
    for k in sorted(acc):
        out.append("  {:<84s} {}".format(k, "  ".join("{}={:.5g}".format(c, v[0]) for c, v in sorted(acc[k].items()))))
    return out


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        print("==", os.path.relpath(path, root))
        if "stats" in os.path.basename(os.path.dirname(path)):
            print("\n".join(kernel_stats(db)))
        else:
            print("\n".join(counters(db)))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
