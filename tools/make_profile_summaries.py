#!/usr/bin/env python3
"""Turn the rocprofv3 databases of one round into the tracked summaries under profiles/.

  python tools/make_profile_summaries.py r01c
reads  gpurun_out/prof_<tag>/bench_results.db, gpurun_out/pmc_<tag>_{sq,fetch,write}/pmc_results.db, gpurun_out/bench_<tag>.json
writes profiles/<tag>_kernel_stats.md, <tag>_pmc.md, <tag>_traffic.json, <tag>_bench.json
"""
import json, os, re, shutil, sqlite3, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")


def keep(line):
    return not any(w in line for w in ("at::native", "rocclr", "nccl"))


out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "rocpd_summary.py"), os.path.join(G, f"prof_{tag}", "bench_results.db")]).decode()
open(os.path.join(P, f"{tag}_kernel_stats.md"), "w").write("\n".join(l for l in out.splitlines() if keep(l)) + "\n")
dbs = [os.path.join(G, f"pmc_{tag}_{k}", "pmc_results.db") for k in ("sq", "fetch", "write")]
out = subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "rocpd_pmc_summary.py")] + dbs).decode()
open(os.path.join(P, f"{tag}_pmc.md"), "w").write("\n".join(l for l in out.splitlines() if keep(l)) + "\n")


def per_kernel(path, counter):
    con = sqlite3.connect(path); per = defaultdict(float); meta = {}; acc = defaultdict(list)
    for did, k, v in con.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name=?", (counter,)):
        per[did] += v; meta[did] = k
    for did, v in per.items():
        acc[meta[did]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}


f = per_kernel(dbs[1], "FETCH_SIZE"); w = per_kernel(dbs[2], "WRITE_SIZE")
tr = {"_how": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --no-cpu-baseline --steps 5`; bytes per launch = FETCH_SIZE[KB] * 1024 * 2 "
              "(gfx950 reports half the bytes of 16 B/lane streaming reads, MI355X_MICROARCH.md) + WRITE_SIZE[KB] * 1024 (uncalibrated for writes; within 10 % of the algorithmic "
              "bytes for the coalesced record writers)"}
for k in f:
    m = re.search(r"(\w+_kernel)", k)
    if m and "qmk" in k:
        tr[m.group(1)] = {"fetch_bytes": f[k] * 2048, "write_bytes": w.get(k, 0) * 1024, "bytes": f[k] * 2048 + w.get(k, 0) * 1024}
json.dump(tr, open(os.path.join(P, f"{tag}_traffic.json"), "w"), indent=1)
shutil.copy(os.path.join(G, f"bench_{tag}.json"), os.path.join(P, f"{tag}_bench.json"))
print(open(os.path.join(P, f"{tag}_kernel_stats.md")).read())
print({k: (round(v["fetch_bytes"] / 1e6), round(v["write_bytes"] / 1e6)) for k, v in tr.items() if k != "_how"})
