"""Per-launch durations of the LU-SGS sweep kernels from a rocprofv3 --kernel-trace database (rocpd sqlite)."""
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select name, start, end from kernels order by start"))
for key in (sys.argv[2:] or ['k_lu_lower_wave', 'k_lu_upper_wave']):
    lu = [(s, e) for n, s, e in rows if key in n]
    if not lu:
        continue
    d = [(e - s) / 1000. for s, e in lu]
    print("%s: %d launches, duration us min %.1f max %.1f mean %.1f, total ms %.2f" % (key, len(d), min(d), max(d), sum(d) / len(d), sum(d) / 1000.))
for r in c.execute("select name, total_calls, total_duration, average from top_kernels limit 8"):
    print("%-70s calls %6d total %.0f us avg %.1f us" % (r[0][:70], r[1], r[2], r[3]))

# duration against the number of workgroups of the launch
import collections
for key in (sys.argv[2:] or ['k_lu_lower_wave', 'k_lu_upper_wave']):
    rows2 = [(gx // max(wx, 1) * (gy // max(wy, 1)) * max(gz, 1), (e - s) / 1000.) for n, s, e, gx, gy, gz, wx, wy in
             c.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x, workgroup_y from kernels order by start") if key in n]
    bins = collections.defaultdict(list)
    for nwg, d in rows2:
        b = 1
        while b < nwg:
            b *= 2
        bins[b].append(d)
    print(key, "workgroups (<=) : mean us (launches)")
    for b in sorted(bins):
        print("   %6d : %6.1f (%d)" % (b, sum(bins[b]) / len(bins[b]), len(bins[b])))
