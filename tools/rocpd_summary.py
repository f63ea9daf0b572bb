"""Per-kernel summary of a rocprofv3 rocpd database (kernel-trace): calls, average / total duration, and how
much the launches overlapped (sum of durations against the union of their intervals)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
t0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0  # skip the first t0 fraction of the trace (warm-up)
rows = list(c.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join "
                      "rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
if not rows:
    sys.exit("no kernel dispatches")
lo, hi = rows[0][1], max(r[2] for r in rows)
cut = lo + (hi - lo) * t0
rows = [r for r in rows if r[1] >= cut]
agg = {}
for name, s, e in rows:
    name = name.split("(")[0][:48]
    a = agg.setdefault(name, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
union, cur_s, cur_e = 0.0, None, None
for _, s, e in rows:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            union += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
union += cur_e - cur_s
total = sum(a[1] for a in agg.values())
for name, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-48s %7d calls %9.1f us avg %11.0f us total" % (name, n, us / n, us))
span = (max(r[2] for r in rows) - rows[0][1]) / 1e3
print("sum of durations %.0f us, union of intervals %.0f us (overlap x%.2f), span %.0f us (GPU busy %.0f %%)" % (
    total, union / 1e3, total / (union / 1e3), span, 100 * union / 1e3 / span))
