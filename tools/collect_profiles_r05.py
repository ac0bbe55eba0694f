#!/usr/bin/env python3
"""Copies the judged summaries of the round-5 evidence run (gpurun_out/r05ev, tools/refresh_profiles_r05.sh) into profiles/."""
import csv
import datetime
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "r05ev")
DST = os.path.join(ROOT, "profiles")
today = datetime.date.today().isoformat()


def have(*p):
    return os.path.exists(os.path.join(SRC, *p))


def copy(src, dst):
    if have(src) and os.path.getsize(os.path.join(SRC, src)) > 0:
        shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, dst))
        print("copied", dst)


names = {"bench_default.json": "r05_sv16_l40_b1024_bench.json", "bench_b64.json": "r05_sv16_l40_b64_bench.json",
         "bench_dense_route.json": "r05_sv16_l40_b1024_bench_dense_route.json",
         "bench_full_size_stages.json": "r05_sv16_l40_b1024_bench_sparse_route_full_size_stages.json",
         "bench_vdag_by_stages.json": "r05_sv16_l40_b1024_bench_projected_sweep_vdag_by_stages.json",
         "bench_under_rocprof.json": "r05_sv16_l40_b1024_bench_under_rocprof.json",
         "bench_cfg4_driver.json": "r05_bench_cfg4_driver.json", "bench_cfg4_under_rocprof.json": "r05_cfg4_driver_bench_under_rocprof.json"}
for name, tag in names.items():
    copy(name, tag)
copy(os.path.join("kt", "kt_kernel_stats.csv"), "r05_sv16_l40_b1024_kernel_stats.csv")
copy(os.path.join("kt_cfg4", "kt_kernel_stats.csv"), "r05_cfg4_driver_kernel_stats.csv")
if have("gpu_tests.log"):
    with open(os.path.join(SRC, "gpu_tests.log")) as f, open(os.path.join(DST, "r05_gpu_tests_tail.txt"), "w") as g:
        g.write("".join(f.readlines()[-4:]))

if have("pmc_f", "f_counter_collection.csv") and have("pmc_w", "w_counter_collection.csv"):
    out = os.path.join(DST, "r05_sv16_l40_b1024_pmc_traffic.json")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(SRC, "pmc_f", "f_counter_collection.csv"),
                    os.path.join(SRC, "pmc_w", "w_counter_collection.csv"), out, "sv16_l40", "1024"], check=True, stdout=subprocess.DEVNULL)
    d = json.load(open(out))
    d["date"] = today
    d["source"] = ("builder-run: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of tools/prof_run5.py (the bench's evaluation: objective_launch on "
                   "the sparse route, objective by projection), tools/refresh_profiles_r05.sh; per kernel NAME, mean over its launches: "
                   "project_staged_kernel = the projection of the target (one launch per evaluation), project_kernel<4, 1> = the lhs tile of "
                   "(later stages)^H y (one launch), apply<12, true> = four launches (psi, the virtual pattern forwards, the virtual z backwards, "
                   "V^H's last stage on the lhs tiles), sweep<12, true> = two (first stage on the lhs tiles, virtual stage); AQC_PROJECTED=0 "
                   "gives the names of the full-size route: apply<12, false> = V^H stage 0, sweep<12, false, false, true> = sweep stage 1")
    json.dump(d, open(out, "w"), indent=1)
    shutil.copyfile(out, os.path.join(DST, "pmc_traffic.json"))
    print("wrote pmc traffic")

sq = [os.path.join(SRC, d, "s_counter_collection.csv") for d in ("sq1", "sq2", "sq3") if have(d, "s_counter_collection.csv")]
if sq:
    txt = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sq_summary.py")] + sq, check=True, capture_output=True, text=True).stdout
    with open(os.path.join(DST, "r05_sq_counters.txt"), "w") as f:
        f.write("# rocprofv3 --pmc passes of tools/prof_run5.py (16 qubits, 40 blocks, 1024 lanes, sparse + projected route, objective by projection; mean per launch of a kernel name), " + today + "\n")
        f.write("# SIMD-cycles of a launch = duration x clock x 1024 SIMDs; GRBM_GUI_ACTIVE / 8 XCDs / duration = the clock the launch really ran at\n")
        f.write(txt)
    print("wrote sq counters")

# config-4 driver: share of the timed run's wall time spent inside kernels (kernel trace of the same command)
if have("kt_cfg4", "kt_kernel_trace.csv") and have("bench_cfg4_under_rocprof.json") and os.path.getsize(os.path.join(SRC, "bench_cfg4_under_rocprof.json")) > 0:
    d = json.load(open(os.path.join(SRC, "bench_cfg4_under_rocprof.json")))
    rows = list(csv.DictReader(open(os.path.join(SRC, "kt_cfg4", "kt_kernel_trace.csv"))))
    st = [int(r["Start_Timestamp"]) for r in rows]
    en = [int(r["End_Timestamp"]) for r in rows]
    win = d["ms_per_step"] * 1e6 * d["steps"]
    t1 = max(en)
    busy = sum(e - s for s, e in zip(st, en) if s >= t1 - win)
    with open(os.path.join(DST, "r05_cfg4_driver_kernel_share.txt"), "w") as f:
        f.write(f"# bench.py --workload cfg4_driver under rocprofv3 --kernel-trace, {today}\n")
        f.write(f"timed run: {d['ms_per_step'] / 1e3:.1f} s wall ({d['value']:.0f} evals/s, {d['config']['horizons']} horizons x {d['config']['restarts_per_horizon']} restarts, "
                f"{d['config']['evaluations']} evaluations)\n")
        f.write(f"kernel time inside that window (sum of the durations of the trace's kernels that start in it; the parity replay after the run is inside it too): "
                f"{busy / 1e6:.1f} ms = {busy / win:.3f} of the wall time\n")
    print("wrote cfg4 share", busy / win)
