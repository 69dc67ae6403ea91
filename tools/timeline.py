"""GPU idle time inside the training step, from a rocprofv3 --kernel-trace rocpd database.

  python tools/timeline.py <results.db> [n_tail_kernels]

Takes the kernels of the tail of the trace (the timed steps), merges their [start, end) intervals over all streams and reports
busy / idle time, the idle gaps by size class, and the largest gaps with the kernels on either side."""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
    print("columns:", cols)
    rows = list(con.execute("select name, start, end from kernels order by start"))
    steps_marker = "adam_tiles" if any("adam_tiles" in r[0] for r in rows) else "adam_multi"
    idx = [i for i, r in enumerate(rows) if steps_marker in r[0]]
    if len(idx) < 4:
        print("fewer than 4 optimiser launches in the trace")
        return
    lo, hi = idx[-4], idx[-1]                     # three whole steps: (adam .. adam]
    seg = rows[lo + 1:hi + 1]
    t0, t1 = rows[lo][2], rows[hi][2]
    span = (t1 - t0) / 3e6
    busy, gaps = 0, []
    cur_end, last_name = t0, rows[lo][0]
    for name, s, e in seg:
        if s > cur_end:
            gaps.append((s - cur_end, last_name, name))
            busy += e - s
            cur_end, last_name = e, name
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end, last_name = e, name
    idle = sum(g[0] for g in gaps)
    print(f"step span {span:.3f} ms; busy {busy / 3e6:.3f} ms, idle {idle / 3e6:.3f} ms per step; {len(seg) / 3:.0f} kernels per step")
    for lim in (2000, 5000, 10000, 50000, 10 ** 9):
        sel = [g for g in gaps if g[0] < lim]
        print(f"  gaps < {lim / 1e3:.0f} us: {len(sel) / 3:.0f} per step, {sum(g[0] for g in sel) / 3e6:.3f} ms per step")
    import collections
    pair = collections.defaultdict(lambda: [0, 0])
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "")[:60]      # noqa: E731
    for g in gaps:
        k = (short(g[1])[:34], short(g[2])[:34])
        pair[k][0] += 1
        pair[k][1] += g[0]
    print("  by (after, before): count per step, us per step")
    for k, v in sorted(pair.items(), key=lambda kv: -kv[1][1])[:14]:
        print(f"    {v[0] / 3:5.1f}  {v[1] / 3e3:7.1f} us   {k[0]}  ->  {k[1]}")
    short = lambda n: n.replace("_ZN12_GLOBAL__N_1", "").replace("(anonymous namespace)::", "")[:60]      # noqa: E731
    for g in sorted(gaps, key=lambda g: -g[0])[:25]:
        print(f"  {g[0] / 1e3:8.1f} us  after {short(g[1])}  before {short(g[2])}")


main()
