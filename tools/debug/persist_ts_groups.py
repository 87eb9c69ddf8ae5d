"""Condenses the per-workgroup trace a -DMHX_PERSIST_TIMING library prints (tools/debug/
persist_ts_timing.py > file): per chain group the master's controller / wait cycles and the
sweep workgroups' poll / vote / sweep cycles per round, when each group ended, and how many
CUs held two workgroups of the launch at the same time."""
import collections
import sys

lines = [l for l in open(sys.argv[1]).read().splitlines() if l.startswith("trace wg")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 400
recs = []
for l in lines[-n:]:
    t = l.split()
    recs.append(dict(x=int(t[2]), y=int(t[3]), cu=(int(t[5]), int(t[7]), int(t[9])), n=int(t[11]),
                     a=int(t[13]), b=int(t[15]), c=int(t[17]), end=int(t[19])))
e0 = min(r["end"] for r in recs)
for y in sorted({r["y"] for r in recs}):
    g = [r for r in recs if r["y"] == y]
    m, sl = g[0], g[1:]
    k = max(len(sl), 1)
    print("group %d: master controller %d wait %d, ended %+.1f us; sweeps: poll %.0f vote %.0f sweep %.0f (max %d)"
          % (y, m["a"], m["b"], (m["end"] - e0) / 100.0, sum(r["a"] for r in sl) / k,
             sum(r["b"] for r in sl) / k, sum(r["c"] for r in sl) / k, max([r["c"] for r in sl] or [0])))
