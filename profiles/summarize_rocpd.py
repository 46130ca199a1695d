#!/usr/bin/env python3
"""Turn rocprofv3's rocpd SQLite output into the small CSV summaries committed under profiles/.

    python profiles/summarize_rocpd.py stats <results.db> <out.csv>          # --kernel-trace --stats
    python profiles/summarize_rocpd.py pmc   <results.db> <out.csv>          # --pmc COUNTER pass
    python profiles/summarize_rocpd.py trace <results.db> <out.csv> [last_n] # --kernel-trace: the last n dispatches in order
    python profiles/summarize_rocpd.py gaps  <results.db> <out.csv> [last_n] # idle time between the last n dispatches, by the kernel that follows the gap
"""
import csv
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    m = re.match(r'([\w:]+(?:<[^()]*>)?)', name)
    return (m.group(1) if m else name)[:110]


def main():
    mode, db, out = sys.argv[1:4]
    cur = sqlite3.connect(db).cursor()
    with open(out, 'w', newline='') as f:
        w = csv.writer(f)
        if mode == 'stats':
            w.writerow(['kernel', 'calls', 'total_us', 'avg_us', 'percent'])
            for name, calls, total, avg, pct in cur.execute(
                    'select name, total_calls, total_duration, average, percentage from top_kernels'):
                w.writerow([short(name), calls, '%.1f' % (total), '%.2f' % (avg), '%.2f' % pct])
        elif mode == 'gaps':
            n = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
            rows = list(cur.execute('select start, end, name from kernels order by start'))[-n:]
            span = rows[-1][1] - rows[0][0]
            busy = sum(e - st for st, e, _ in rows)
            by = {}
            for (st0, e0, n0), (st1, e1, n1) in zip(rows, rows[1:]):
                g = max(0, st1 - e0)
                k = short(n1)
                a = by.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += g
            w.writerow(['span_us %.1f busy_us %.1f idle_us %.1f over %d dispatches' % (span / 1e3, busy / 1e3, (span - busy) / 1e3, len(rows)), '', ''])
            w.writerow(['kernel_after_gap', 'gaps', 'idle_us'])
            for k, (c, g) in sorted(by.items(), key=lambda kv: -kv[1][1])[:40]:
                w.writerow([k, c, '%.1f' % (g / 1e3)])
        elif mode == 'trace':
            n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
            w.writerow(['start_us', 'duration_us', 'grid', 'workgroup', 'kernel'])
            cols = [d[0] for d in cur.execute('select * from kernels limit 1').description]
            grid = 'grid_size' if 'grid_size' in cols else 'grid_x' if 'grid_x' in cols else 'grid_size_x' if 'grid_size_x' in cols else '0'
            wgs = 'workgroup_size' if 'workgroup_size' in cols else 'workgroup_x' if 'workgroup_x' in cols else 'workgroup_size_x' if 'workgroup_size_x' in cols else '0'
            rows = list(cur.execute('select start, end - start, %s, %s, name from kernels order by start' % (grid, wgs)))[-n:]
            t0 = rows[0][0] if rows else 0
            for st, dur, g, wg, name in rows:
                w.writerow(['%.1f' % ((st - t0) / 1e3), '%.1f' % (dur / 1e3), g, wg, short(name)])
        else:
            w.writerow(['kernel', 'counter', 'dispatches', 'avg_value_per_dispatch'])
            for name, ctr, n, avg in cur.execute(
                    'select kernel_name, counter_name, count(*), avg(value) from counters_collection '
                    'group by kernel_name, counter_name order by avg(value) desc'):
                if avg and avg > 1000:
                    w.writerow([short(name), ctr, n, '%.1f' % avg])


if __name__ == '__main__':
    main()
