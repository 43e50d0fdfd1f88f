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
    for name, n, tot, avg, mn, mx, vg, ag, sg, lds, grid, wg in rows[:30]:
        out.append("  {:<84s} calls {:>4d}  avg_us {:>10.2f}  min_us {:>10.2f}  max_us {:>10.2f}  pct {:>5.1f}  vgpr {:>3} agpr {:>3} sgpr {:>3} lds {:>6} grid_threads {:>8} wg {}".format(
            short(name), n, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, ag, sg, lds, grid, wg))
    return out


def timeline(db, last=14):
    """The last dispatches of the trace in start order: start relative to the first of them, duration, and the gap between the end
    of the previous dispatch and this one's start (launch gaps between the dependent kernels of one head call)."""
    cur = db.cursor()
    rows = cur.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                       "on d.kernel_id = s.id order by d.start desc limit ?", (last,)).fetchall()[::-1]
    if not rows:
        return []
    t0, prev_end, out, gaps = rows[0][1], None, [], 0.0
    for name, a, b in rows:
        gap = (a - prev_end) / 1e3 if prev_end is not None else 0.0
        gaps += max(gap, 0.0)
        out.append("  t={:>9.2f} us  dur {:>9.2f} us  gap {:>7.2f} us  {}".format((a - t0) / 1e3, (b - a) / 1e3, gap, short(name)[:70]))
        prev_end = b
    out.append("  span {:.2f} us, of which gaps {:.2f} us".format((rows[-1][2] - t0) / 1e3, gaps))
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


CONV1_LIKE = ("%conv_mfma_kernelILi7%", "%conv_f16x3_kernelILi7%")


def conv1_traffic(root, classes, out_path, like=CONV1_LIKE, label="TransformNet conv 7x7 (conv_mfma_kernel<7,..> or conv_f16x3_kernel<7,..>)"):
    """FETCH_SIZE / WRITE_SIZE (KiB per dispatch, separate passes) of one kernel (default: the conv 7x7 kernel; --kernel
    <sql like pattern> for another, e.g. %spectral_gemm_kernelILi2%) -> bytes per launch / class.
    gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at half their bytes."""
    import json
    where = "(" + " or ".join("s.kernel_name like '{}'".format(k) for k in like) + ")"
    vals = {}
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        for path in glob.glob(os.path.join(root, "pmc_" + name, "*.db")):
            cur = sqlite3.connect(path).cursor()
            row = cur.execute(
                "select avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                "where p.name = ? and " + where, (name,)).fetchone()
            vals[name] = row[0]
    if len(vals) == 2 and all(v is not None for v in vals.values()):
        total = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
        # matrix-pipe utilisation of the same kernel from the SQ pass: SQ_VALU_MFMA_BUSY_CYCLES counts units of 32 cycles
        # summed over the SIMDs (= one per 32x32x16 f16 MFMA, two per 32x32x2 f32 MFMA); 1024 SIMDs on the chip
        busy = {}
        for path in glob.glob(os.path.join(root, "pmc_sq", "*.db")):
            cur = sqlite3.connect(path).cursor()
            row = cur.execute(
                "select avg(d.end - d.start) from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                "where " + where).fetchone()
            busy["avg_launch_us"] = row[0] / 1e3 if row and row[0] else None     # duration in the SAME (profiled) pass
            for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
                row = cur.execute(
                    "select avg(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id "
                    "join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id "
                    "where p.name = ? and " + where, (name,)).fetchone()
                busy[name] = row[0]
        with open(out_path, "w") as f:
            json.dump({"kernel": label, "classes_profiled": classes, "fetch_kib": vals["FETCH_SIZE"],
                       "write_kib": vals["WRITE_SIZE"], "fetch_correction": 2.0, "bytes_per_launch": total,
                       "bytes_per_class": total / classes, "source": os.path.basename(os.path.normpath(root)),
                       "mfma_busy_cycles_x32": busy.get("SQ_VALU_MFMA_BUSY_CYCLES"), "grbm_gui_active": busy.get("GRBM_GUI_ACTIVE"),
                       "avg_launch_us": busy.get("avg_launch_us"),
                       "effective_clock_ghz": (round(busy["GRBM_GUI_ACTIVE"] / busy["avg_launch_us"] / 1e3, 3)
                                               if busy.get("GRBM_GUI_ACTIVE") and busy.get("avg_launch_us") else None),
                       "mfma_pipe_busy": (round(busy["SQ_VALU_MFMA_BUSY_CYCLES"] * 32 / (1024 * busy["GRBM_GUI_ACTIVE"]), 4)
                                          if busy.get("SQ_VALU_MFMA_BUSY_CYCLES") and busy.get("GRBM_GUI_ACTIVE") else None)}, f, indent=1)
        print("{} traffic: {:.1f} MB per launch ({} classes) -> {}".format(label, total / 1e6, classes, out_path))


def main(root):
    for path in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        db = sqlite3.connect(path)
        print("==", os.path.relpath(path, root))
        if "stats" in os.path.basename(os.path.dirname(path)):
            print("\n".join(kernel_stats(db)))
            print("== timeline of the last dispatches")
            print("\n".join(timeline(db)))
        else:
            print("\n".join(counters(db)))


if __name__ == "__main__":
    if "--traffic" in sys.argv:       # summarize_prof.py <dir> --traffic <classes> <out.json> [--kernel <like pattern>]
        i = sys.argv.index("--traffic")
        if "--kernel" in sys.argv:
            k = sys.argv[sys.argv.index("--kernel") + 1]
            conv1_traffic(sys.argv[1], int(sys.argv[i + 1]), sys.argv[i + 2], like=(k,), label=k.strip("%"))
        else:
            conv1_traffic(sys.argv[1], int(sys.argv[i + 1]), sys.argv[i + 2])
    else:
        main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out")
