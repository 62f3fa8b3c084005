#!/usr/bin/env python3
"""Concurrency / phase breakdown of one bench step from a rocprofv3 kernel trace (rocpd sqlite db or kernel_trace.csv).
usage: timeline.py results.db [step_index_from_end] [--list]   (--list: every kernel of the step: start offset, duration, queue)"""
import csv
import sqlite3
import sys


def load(path):
    if path.endswith(".csv"):
        rows = []
        for r in csv.DictReader(open(path)):
            rows.append((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
        return rows
    db = sqlite3.connect(path)
    cols = [c[1] for c in db.execute("pragma table_info(kernels)")]
    q = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
    return [(n, s, e, str(qq), str(qq)) for n, s, e, qq in db.execute(f"select name, start, end, {q} from kernels order by start")]


def main():
    want_list = "--list" in sys.argv
    if want_list:
        sys.argv.remove("--list")
    rows = sorted(load(sys.argv[1]), key=lambda r: r[1])
    # a step ends with k_dual_step; take the window between the last two (or chosen) occurrences
    marks = [i for i, r in enumerate(rows) if r[0].startswith("k_dual_step")]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    lo, hi = marks[-k - 1] + 1, marks[-k] + 1
    step = rows[lo:hi]
    t0, t1 = step[0][1], max(r[2] for r in step)
    print(f"step window: {len(step)} kernels, {(t1 - t0) / 1e6:.3f} ms")
    # sweep: time with 0 / 1 / >=2 kernels in flight
    ev = []
    for n, s, e, q, st in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    cur, last, acc = 0, t0, {0: 0, 1: 0, 2: 0}
    for t, d in ev:
        acc[min(cur, 2)] += t - last
        last = t
        cur += d
    tot = t1 - t0
    print("GPU idle %.2f ms (%.1f%%), one kernel %.2f ms (%.1f%%), two or more %.2f ms (%.1f%%)" % (
        acc[0] / 1e6, 100 * acc[0] / tot, acc[1] / 1e6, 100 * acc[1] / tot, acc[2] / 1e6, 100 * acc[2] / tot))
    # per-queue busy time
    byq = {}
    for n, s, e, q, st in step:
        byq.setdefault(q, []).append((s, e, n))
    for q, lst in byq.items():
        busy = sum(e - s for s, e, _ in lst)
        print(f"queue {q}: {len(lst)} kernels, busy {busy / 1e6:.2f} ms, span {(max(e for _, e, _ in lst) - min(s for s, _, _ in lst)) / 1e6:.2f} ms, first {(lst[0][0] - t0) / 1e6:.2f} ms")
    # per-queue gaps (time the queue has nothing running inside its span) and the kernels that follow the largest ones
    for q, lst in byq.items():
        lst = sorted(lst)
        gaps_q, end = [], lst[0][0]
        for s, e, n in lst:
            if s > end:
                gaps_q.append((s - end, (end - t0) / 1e6, n[:60]))
            end = max(end, e)
        gaps_q.sort(reverse=True)
        print(f"  queue {q}: {sum(g for g, _, _ in gaps_q) / 1e6:.2f} ms of gaps in {len(gaps_q)} gaps; largest (us @ ms -> next kernel): " +
              "; ".join("%.0f @ %.2f -> %s" % (g / 1e3, at, n) for g, at, n in gaps_q[:6]))
    # phase markers
    def first(pred):
        for n, s, e, q, st in step:
            if pred(n):
                return (s - t0) / 1e6
        return -1
    print("first loss kernel at %.2f ms; first adamw at %.2f ms; dual_step at %.2f ms" % (
        first(lambda n: "k_loss" in n), first(lambda n: "k_adamw" in n), first(lambda n: n.startswith("k_dual_step"))))
    # biggest idle gaps
    gaps = []
    cur, last = 0, t0
    for t, d in ev:
        if cur == 0 and t > last:
            gaps.append((t - last, (last - t0) / 1e6))
        last = t
        cur += d
    gaps.sort(reverse=True)
    print("largest idle gaps (us @ ms):", [(round(g / 1e3, 1), round(at, 2)) for g, at in gaps[:8]])
    if want_list:
        qs = {q: i for i, q in enumerate(sorted(byq))}
        print("start_us dur_us queue kernel")
        for n, s, e, q, st in step:
            print("%9.1f %7.1f  q%d %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, qs[q], "    " * qs[q], n[:70]))


if __name__ == "__main__":
    main()
