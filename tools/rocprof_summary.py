#!/usr/bin/env python3
"""Aggregate a rocprofv3 rocpd database (…_results.db from `rocprofv3 --kernel-trace --stats`) into
the per-kernel CSV kept under profiles/.   usage: rocprof_summary.py results.db out.csv "header comment" """
import sqlite3
import sys


def main(db_path, out, comment=""):
    db = sqlite3.connect(db_path)
    rows = db.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                      "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    with open(out, "w") as f:
        for line in comment.split("\\n"):
            if line:
                f.write("# " + line + "\n")
        f.write("name,calls,total_ns,avg_ns,min_ns,max_ns,percent\n")
        for r in rows:
            f.write('"%s",%d,%d,%.1f,%d,%d,%.2f\n' % (r[0].replace('"', "'"), r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot))
    print(f"{out}: {len(rows)} kernels, {tot/1e6:.2f} ms total")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else "")
