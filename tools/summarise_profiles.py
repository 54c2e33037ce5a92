#!/usr/bin/env python
"""Turn one tools/gpu_round.sh output directory (rocprofv3 rocpd databases) into the tracked
summaries under profiles/:

    python tools/summarise_profiles.py gpurun_out/<tag> <name>

writes  profiles/<name>_kernel_stats.csv   per-kernel calls / total / average / min / max (ns) of the
                                            `rocprofv3 --kernel-trace --stats` run
        profiles/<name>_pmc.json           per-launch medians of every counter of the --pmc passes
                                            for the dominant kernel, and derived figures
        profiles/pmc_latest.json           what bench.py reads for `roofline.traffic`
HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md section "HBM": FETCH_SIZE / WRITE_SIZE are
in KB; on gfx950 FETCH_SIZE tallies 128-B requests of wide coalesced reads at 64 B, so the upper
bound of the read traffic is 2 x FETCH_SIZE.  Both bounds are stored; `hbm_bytes_per_launch` is the
corrected (upper) one.
"""
import glob
import json
import os
import sqlite3
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_rows(db_path):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                      "from kernels group by name order by sum(duration) desc").fetchall()
    regs = {r[0]: r[1:] for r in db.execute(
        "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), "
        "max(scratch_size) from kernels group by name")}
    return rows, regs


def step_launch_filter(db, kernel_like):
    """dispatch ids of the dominant kernel's step launches = the launches with the modal duration
    class (start / reset launches run one dynamics evaluation instead of four and are shorter)."""
    rows = db.execute("select dispatch_id, duration from kernels where name like ?", (kernel_like,)).fetchall()
    if not rows:
        return set()
    med = statistics.median(d for _, d in rows)
    return {i for i, d in rows if d > 0.6 * med}


def summarise_kernel(src, stats_db, like):
    """Step-launch durations and per-launch counter medians of one kernel."""
    dominant = like
    db = sqlite3.connect(stats_db)
    keep = step_launch_filter(db, like)
    durs = [d for i, d in db.execute("select dispatch_id, duration from kernels where name like ?", (like,))
            if i in keep]
    summary = {"source": os.path.relpath(os.path.abspath(src), ROOT), "dominant_kernel": dominant,
               "step_launches": len(durs), "avg_step_launch_ns": statistics.mean(durs),
               "median_step_launch_ns": statistics.median(durs), "counters_per_launch_median": {}}
    for p in sorted(glob.glob(os.path.join(src, "pmc_*", "*.db"))):
        d = sqlite3.connect(p)
        med = statistics.median(x for (x,) in d.execute(
            "select duration from kernels where name like ?", (like,)))
        vals = {}
        for cname, value, dur in d.execute(
                "select counter_name, value, duration from counters_collection where kernel_name like ?", (like,)):
            if dur > 0.6 * med:
                vals.setdefault(cname, []).append(value)
        for k, v in vals.items():
            summary["counters_per_launch_median"][k] = statistics.median(v)
            summary.setdefault("pass_median_ns", {})[k] = med     # (a launch under counters is 3-6 % longer than without)
    c = summary["counters_per_launch_median"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        lo = (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        hi = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        summary["hbm_bytes_per_launch_uncorrected"] = lo
        summary["hbm_bytes_per_launch"] = hi
    if "SQ_WAVES" in c and c["SQ_WAVES"]:
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"):
            if k in c:
                summary[k + "_per_wave"] = c[k] / c["SQ_WAVES"]
    if c.get("SQ_WAVE_CYCLES"):
        for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
            if k in c:
                summary[k + "_frac_of_wave_cycles"] = c[k] / c["SQ_WAVE_CYCLES"]
    # clock the chip sustained under this kernel: GRBM_GUI_ACTIVE counts busy gfx-clock cycles of every XCD (8 on the
    # MI355X; calibrated in round 5 against the s_memtime / s_memrealtime clock of tools/microbench/op_issue) over the
    # dispatch -- taken against the median duration of the pass that collected it
    if c.get("GRBM_GUI_ACTIVE") and summary.get("pass_median_ns", {}).get("GRBM_GUI_ACTIVE"):
        summary["sustained_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / summary["pass_median_ns"]["GRBM_GUI_ACTIVE"]
    # measured occupancy of the VALU issue pipes: SQ_ACTIVE_INST_VALU is in quad-cycles summed over the SIMDs
    if c.get("SQ_ACTIVE_INST_VALU") and summary.get("pass_median_ns", {}).get("SQ_ACTIVE_INST_VALU"):
        clk = summary.get("sustained_clock_ghz", 2.4)
        summary["valu_pipe_busy_frac"] = 4.0 * c["SQ_ACTIVE_INST_VALU"] / 1024.0 / (clk * summary["pass_median_ns"]["SQ_ACTIVE_INST_VALU"])
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        summary["l2_hit_rate"] = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
    return summary


def main():
    src, name = sys.argv[1], sys.argv[2]
    # optional third argument: output directory (the GPU box summarises in place, the rocpd
    # databases themselves are too large to travel back)
    out_dir = sys.argv[3] if len(sys.argv) > 3 else os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    stats_db = glob.glob(os.path.join(src, "stats", "*.db"))[0]
    rows, regs = kernel_rows(stats_db)
    total = sum(r[2] for r in rows)
    with open(os.path.join(out_dir, f"{name}_kernel_stats.csv"), "w") as f:
        f.write("kernel,calls,total_ns,avg_ns,min_ns,max_ns,pct,arch_vgpr,accum_vgpr,sgpr,lds_bytes,scratch_bytes\n")
        for r in rows:
            g = regs.get(r[0], ("",) * 5)
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f,%s,%s,%s,%s,%s\n' % (r[0], r[1], r[2], r[3], r[4], r[5],
                                                                   100.0 * r[2] / total, *g))
    # JM_PROFILE_DOMINANT: LIKE pattern of the kernel to summarise instead of the one with the largest total
    # (launches made of several kernels: the reset launches of a short run can outweigh the step kernels);
    # JM_PROFILE_KERNELS: comma-separated LIKE patterns of further kernels summarised under "other_kernels".
    dominant = rows[0][0]
    pat = os.environ.get("JM_PROFILE_DOMINANT")
    if pat:
        dominant = next((r[0] for r in rows if sqlite3.connect(":memory:").execute("select ? like ?", (r[0], pat)).fetchone()[0]), dominant)
    summary = summarise_kernel(src, stats_db, dominant)
    c = summary["counters_per_launch_median"]
    others = [x for x in os.environ.get("JM_PROFILE_KERNELS", "").split(",") if x]
    if others:
        summary["other_kernels"] = {}
        for o in others:
            name_o = next((r[0] for r in rows if sqlite3.connect(":memory:").execute("select ? like ?", (r[0], o)).fetchone()[0]), None)
            if name_o:
                summary["other_kernels"][name_o] = summarise_kernel(src, stats_db, name_o)
    try:
        with open(os.path.join(src, "bench.json")) as f:
            b = json.loads(f.read().strip().splitlines()[-1])
        summary["bench_line"] = b
        summary["model"], summary["batch"], summary["dtype"] = (
            b["config"]["workload"].split()[0], b["config"]["lanes_per_gpu"], b["dtype"])
        summary["extra_terms"] = b["config"].get("extra_terms", "sensors")
    except Exception as e:  # noqa: BLE001
        summary["bench_line"] = f"unavailable: {e}"
    with open(os.path.join(out_dir, f"{name}_pmc.json"), "w") as f:
        json.dump(summary, f, indent=1)
    latest = {k: summary.get(k) for k in ("model", "batch", "dtype", "extra_terms", "hbm_bytes_per_launch",
                                          "hbm_bytes_per_launch_uncorrected", "dominant_kernel", "source")}
    latest["valu_insts_per_wave"] = summary.get("SQ_INSTS_VALU_per_wave")
    latest["waves_per_launch"] = c.get("SQ_WAVES")
    latest["sustained_clock_ghz"] = summary.get("sustained_clock_ghz")
    latest["valu_pipe_busy_frac"] = summary.get("valu_pipe_busy_frac")
    latest["from"] = f"profiles/{name}_pmc.json"
    # fourth argument: name of the "latest" file bench.py reads (pmc_latest.json / pmc_con_latest.json)
    latest_name = sys.argv[4] if len(sys.argv) > 4 else "pmc_latest.json"
    with open(os.path.join(out_dir, latest_name), "w") as f:
        json.dump(latest, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if k != "bench_line"}, indent=1))


if __name__ == "__main__":
    main()
