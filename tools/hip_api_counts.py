"""HIP API calls of a rocprofv3 --hip-trace run (rocpd SQLite output), by name: count and total host time inside the
calls -- over the whole run, or (--window) between the last two hipDeviceSynchronize calls of the process, which is how
tools/batch_cpu_profile.py brackets its timed batch.
usage: python tools/hip_api_counts.py <dir> [divide_by] [--window]"""
import glob, os, sqlite3, sys
dbs = glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True)
div = float(sys.argv[2]) if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else 1.0
window = "--window" in sys.argv
for db in dbs:
    con = sqlite3.connect(db)
    tabs = [r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
    reg = [t for t in tabs if t.startswith("rocpd_region") and "string" not in t][0]
    strs = [t for t in tabs if t.startswith("rocpd_string")][0]
    where = ""
    if window:
        syncs = list(con.execute(f"select r.start, r.end from {reg} r join {strs} s on r.name_id = s.id where s.string = 'hipDeviceSynchronize' order by r.start"))
        lo, hi = syncs[-2][1], syncs[-1][0]
        where = f"where r.start >= {lo} and r.start <= {hi}"
        print(f"window: {(hi - lo) / 1e6:.1f} ms between the last two hipDeviceSynchronize calls")
    q = f"select s.string, count(*), sum(r.end - r.start), count(distinct r.tid) from {reg} r join {strs} s on r.name_id = s.id {where} group by s.string order by 3 desc"
    for name, n, ns, tids in con.execute(q):
        print(f"{name:40s} {n / div:10.1f} calls  {ns / div / 1e3:10.1f} us  ({ns / max(n, 1) / 1e3:.1f} us each, {tids} threads)")
    if "--memcpy" in sys.argv:   # the copies by direction and size (arguments as rocprofv3 recorded them)
        import collections, json
        agg = collections.defaultdict(lambda: [0, 0])
        argt = [t for t in tabs if t.startswith("rocpd_arg")]
        if argt:
            acols = [r[1] for r in con.execute(f"pragma table_info({argt[0]})")]
            print(argt[0], acols)
            q = (f"select r.id, r.end - r.start, a.name, a.value from {reg} r join {strs} s on r.name_id = s.id join {argt[0]} a on a.event_id = r.event_id "
                 f"{where + ' and' if where else 'where'} s.string = 'hipMemcpyAsync' and a.name in ('kind', 'sizeBytes')")
            rows = collections.defaultdict(dict)
            for rid, ns, an, av in con.execute(q):
                rows[rid][an] = av
                rows[rid]["ns"] = ns
            for r in rows.values():
                k = (r.get("kind"), r.get("sizeBytes"))
                agg[k][0] += 1
                agg[k][1] += r["ns"]
            for k, (n, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                print(f"hipMemcpyAsync kind {k[0]} bytes {k[1]}: {n / div:.2f} calls, {ns / div / 1e3:.1f} us ({ns / n / 1e3:.1f} us each)")
