"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per launch).

  python tools/pmc_summary.py hbm  <fetch counter_collection.csv> <write counter_collection.csv>  > profiles/rN/bench_pmc_hbm_bytes.csv
  python tools/pmc_summary.py sq   <sq counter_collection.csv>                                   > profiles/rN/bench_pmc_sq.csv
  python tools/pmc_summary.py stats <results.db>                                                 > profiles/rN/bench_kernel_stats.csv
Inputs may be rocprofv3 CSVs (--output-format csv) or its default rocpd sqlite database (*_results.db).

FETCH_SIZE / WRITE_SIZE are reported in KiB per launch, RAW: on gfx950 FETCH_SIZE under-reports wide coalesced
reads by 2x (MI355X_MICROARCH.md, HBM/rocprofv3 section) -- bench.py doubles it when it quotes `roofline.traffic`.
"""
import collections
import csv
import sys


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(set)
    if path.endswith(".db"):                      # rocprofv3's default rocpd (sqlite) output
        import sqlite3
        con = sqlite3.connect(path)
        for k, cname, v, d in con.execute("select kernel_name, counter_name, value, dispatch_id from counters_collection"):
            acc[k][cname] += float(v)
            launches[k].add(d)
        return acc, {k: len(v) for k, v in launches.items()}
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"]
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in launches.items()}


def kernel_stats(path):
    """the --stats table (per kernel: calls, total / average / min / max duration in ns) from a rocpd database"""
    import sqlite3
    import statistics
    con = sqlite3.connect(path)
    d = collections.defaultdict(list)
    for name, dur in con.execute("select name, duration from kernels"):
        d[name].append(int(dur))
    tot = sum(sum(v) for v in d.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"')
    for name, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        sd = statistics.pstdev(v) if len(v) > 1 else 0.0
        print(f'"{name}",{len(v)},{sum(v)},{sum(v) / len(v):.6f},{100.0 * sum(v) / tot:.2f},{min(v)},{max(v)},{sd:.6f}')


def main():
    mode = sys.argv[1]
    if mode == "stats":
        return kernel_stats(sys.argv[2])
    if mode == "hbm":
        print("counter,kernel,launches,kib_per_launch_raw   (FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950: double it)")
        for path in sys.argv[2:]:
            acc, n = per_kernel(path)
            rows = sorted(((c, k, n[k], v / n[k]) for k, d in acc.items() for c, v in d.items()), key=lambda r: -r[3] * r[2])
            for c, k, ln, v in rows[:24]:
                print(f"{c},{k},{ln},{v:.1f}")
    else:
        acc, n = per_kernel(sys.argv[2])
        names = sorted({c for d in acc.values() for c in d})
        print("kernel,launches," + ",".join(c + "_per_launch" for c in names) + ",mfma_busy_frac")
        order = sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", 0.0))
        for k in order[:24]:
            d = acc[k]
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the sampled SIMDs; with GRBM_GUI_ACTIVE wall cycles
            # the normalisation that makes a back-to-back MFMA loop read 1.0 on this part is active * 128
            busy = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            act = d.get("GRBM_GUI_ACTIVE", 0.0)
            frac = busy / (act * 128) if act else 0.0
            print(k + "," + str(n[k]) + "," + ",".join(f"{d.get(c, 0.0) / n[k]:.0f}" for c in names) + f",{frac:.3f}")


if __name__ == "__main__":
    main()
