#!/usr/bin/env python3
"""Restrict a rocprofv3 kernel trace of `bench.py` to consecutive UVC-train steps and print the per-step table kept under
profiles/*_kernel_stats_uvc_train_steps_only.csv.  A step ends with one k_dual_step launch (uvc_optimizer's dual update), so the
window between the k_dual_step marks FIRST and LAST holds LAST - FIRST whole steps and none of bench.py's stand-alone kernel table.

    usage: rocprof_steps_only.py results.db out.csv [first_mark=5] [last_mark=24]
"""
import sqlite3
import sys


def main(db_path, out, first=5, last=24):
    db = sqlite3.connect(db_path)
    marks = [r[0] for r in db.execute("select end from kernels where name like '%k_dual_step%' order by start").fetchall()]
    if len(marks) <= last:
        raise SystemExit(f"only {len(marks)} k_dual_step launches in the trace")
    t0, t1, nsteps = marks[first], marks[last], last - first
    rows = db.execute("select name, count(*), sum(end-start) from kernels where start >= ? and end <= ? group by name order by 3 desc", (t0, t1)).fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w") as f:
        f.write(f"# same trace, restricted to {nsteps} consecutive UVC-train steps (between k_dual_step marks {first}..{last}): per-kernel launches per step, "
                f"mean duration, ms per step; window {(t1 - t0) / nsteps / 1e6:.3f} ms per step\n")
        f.write("name,calls_per_step,avg_us,ms_per_step,percent\n")
        for name, n, ns in rows:
            f.write('"%s",%.1f,%.1f,%.3f,%.2f\n' % (name.replace('"', "'"), n / nsteps, ns / n / 1e3, ns / nsteps / 1e6, 100.0 * ns / tot))
    print(f"{out}: {len(rows)} kernels, {tot / nsteps / 1e6:.2f} ms of kernel time per step, {(t1 - t0) / nsteps / 1e6:.2f} ms wall per step")


if __name__ == "__main__":
    a = sys.argv
    main(a[1], a[2], int(a[3]) if len(a) > 3 else 5, int(a[4]) if len(a) > 4 else 24)
