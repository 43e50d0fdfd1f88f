#!/usr/bin/env python3
"""HBM traffic counters (FETCH_SIZE x 2 + WRITE_SIZE, separate rocprofv3 PMC passes - bench.live_counters) and launch times of the
step's kernels under the library $OS2D_HIP_LIB names: one line per kernel.  tools/ab_counters.py <classes> [kernel substring ...]"""
import os, sys
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)
import bench
classes = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
want = [a for a in sys.argv[2:] if a != "sq"]
SQ = ("SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_VALU_MFMA_BUSY_CYCLES",
      "GRBM_GUI_ACTIVE")
sq = "sq" in sys.argv[2:]      # "sq": one pass of wave-state counters instead of the two traffic passes
lc = bench.live_counters("fftx3", classes, passes=((SQ,) if sq else (("FETCH_SIZE",), ("WRITE_SIZE",))), timeout_s=240)
lib = os.environ.get("OS2D_HIP_LIB", "product")
lib = lib.split("/")[-2] if "/" in lib else lib
if "error" in lc:
    print("error", lc["error"])
for name, k in sorted(lc.get("kernels", {}).items()):
    if want and not any(w in name for w in want):
        continue
    if sq:
        wc = k.get("SQ_WAVE_CYCLES") or float("nan")
        print("[{} {}] {:28s} {:9.1f} us  wait_any {:.3f}  wait_inst {:.3f}  active {:.3f}  wait_inst_lds {:.3f} of the wave cycles; lds conflicts {:.3g}; "
              "mfma busy {}".format(lib, classes, name, k.get("avg_launch_us_SQ_WAVE_CYCLES", float("nan")), k.get("SQ_WAIT_ANY", 0) / wc,
                                    k.get("SQ_WAIT_INST_ANY", 0) / wc, k.get("SQ_ACTIVE_INST_ANY", 0) / wc, k.get("SQ_WAIT_INST_LDS", 0) / wc,
                                    k.get("SQ_LDS_BANK_CONFLICT", 0), k.get("mfma_pipe_busy")))
        continue
    print("[{} {}] {:28s} launches {:3d}  {:9.1f} us  fetch x2 {:8.1f} MB  write {:8.1f} MB".format(
        lib, classes, name, k.get("launches", 0), k.get("avg_launch_us_FETCH_SIZE", float("nan")),
        2.0 * k.get("FETCH_SIZE", float("nan")) / 1024.0, k.get("WRITE_SIZE", float("nan")) / 1024.0))
