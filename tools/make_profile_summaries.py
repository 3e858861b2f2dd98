#!/usr/bin/env python3
"""Turn the rocprofv3 databases of one round into the tracked summaries under profiles/.

  python tools/make_profile_summaries.py r01c
reads  gpurun_out/prof_<tag>/bench_results.db, gpurun_out/pmc_<tag>_{sq,fetch,write}/pmc_results.db, gpurun_out/bench_<tag>.json
writes profiles/<tag>_kernel_stats.md, <tag>_pmc.md, <tag>_traffic.json, <tag>_bench.json, <tag>_counters.json and profiles/CURRENT (= <tag>)

  python tools/make_profile_summaries.py --counters-only r02m     # only <tag>_counters.json + CURRENT, from the summaries already under profiles/

<tag>_counters.json is what bench.py reads for roofline.traffic / mfma_busy / hbm_gbps (per kernel: HBM bytes per launch, matrix-core busy
fraction, launch duration of the profiled run); profiles/CURRENT names the set, so a bench line can never quote a stale file by name.
Columns of <tag>_kernel_stats.md: "VGPR" is rocprofv3's number, HALF of the unified register allocation of a wave64 kernel on gfx950 (the kernel
descriptor's granulated count, 8-register granules, e.g. lq_node_kernel: .vgpr_count 177 in the assembly metadata -> 184 allocated -> 92 shown);
DESIGN.md quotes the assembly metadata.
"""
import json, os, re, shutil, sqlite3, subprocess, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = [a for a in sys.argv[1:] if not a.startswith("-")][0]
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
CLOCK_HZ, SIMDS = 2.4e9, 1024


def md_table(path):
    rows = [l.strip().strip("|").split("|") for l in open(path) if l.startswith("|") and not l.startswith("|---")]
    hdr = [c.strip() for c in rows[0]]
    return [dict(zip(hdr, (c.strip() for c in r))) for r in rows[1:]]


def write_counters():
    """<tag>_counters.json from <tag>_kernel_stats.md (launch durations), <tag>_pmc.md (SQ_VALU_MFMA_BUSY_CYCLES) and <tag>_traffic.json (HBM bytes)"""
    dur = {re.search(r"(\w+_kernel)", r["kernel"]).group(1): float(r["avg us"]) for r in md_table(os.path.join(P, f"{tag}_kernel_stats.md")) if re.search(r"(\w+_kernel)", r["kernel"])}
    pmc = {re.search(r"(\w+_kernel)", r["kernel"]).group(1): r for r in md_table(os.path.join(P, f"{tag}_pmc.md")) if re.search(r"(\w+_kernel)", r["kernel"])}
    tr = json.load(open(os.path.join(P, f"{tag}_traffic.json")))
    out = {"_how": "per kernel, per launch: bytes = 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes, MI355X_MICROARCH.md corrections); mfma_busy = "
                   "SQ_VALU_MFMA_BUSY_CYCLES / (avg launch duration of the kernel-trace pass x 2.4 GHz x 1024 SIMDs); issue_util = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; "
                   "all from `bench.py --no-cpu-baseline` (configs[1], 256 instances, N = 100)", "_tag": tag}
    for k, d in dur.items():
        e = {"avg_us": d}
        if k in tr and isinstance(tr[k], dict):
            e.update(bytes=tr[k]["bytes"], fetch_bytes=tr[k]["fetch_bytes"], write_bytes=tr[k]["write_bytes"], hbm_gbps=tr[k]["bytes"] / (d * 1e-6) / 1e9)
        if k in pmc:
            r = pmc[k]
            busy = float(r.get("SQ_VALU_MFMA_BUSY_CYCLES") or 0.0)
            e["mfma_busy"] = busy / (d * 1e-6 * CLOCK_HZ * SIMDS)
            if float(r.get("SQ_WAVE_CYCLES") or 0.0) > 0:
                e["issue_util"] = float(r["SQ_ACTIVE_INST_ANY"]) / float(r["SQ_WAVE_CYCLES"])
        out[k] = e
    json.dump(out, open(os.path.join(P, f"{tag}_counters.json"), "w"), indent=1)
    open(os.path.join(P, "CURRENT"), "w").write(tag + "\n")
    print("wrote", f"profiles/{tag}_counters.json", "and profiles/CURRENT;", {k: (round(v.get("mfma_busy", 0), 4), round(v.get("hbm_gbps", 0))) for k, v in out.items() if isinstance(v, dict)})


if "--counters-only" in sys.argv:
    write_counters()
    sys.exit(0)


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
for src, dst in ((f"bench_{tag}_driver.json", f"{tag}_bench_driver.json"), (f"bench_{tag}_driver_no_overlap.json", f"{tag}_bench_driver_no_overlap.json"),
                 (f"bench_{tag}_driver_carry.json", f"{tag}_bench_driver_carry.json"), (f"bench_{tag}_no_overlap.json", f"{tag}_bench_no_overlap.json"),
                 (f"bench_{tag}_sweep.json", f"{tag}_bench_sweep.json"), ("adapter_latency.json", f"{tag}_adapter_latency.json"), (f"pcie_rate_{tag}.txt", f"{tag}_pcie_rate.txt"),
                 (f"wbc_tail_{tag}.txt", f"{tag}_wbc_tail.txt"), (f"timeline_{tag}.txt", f"{tag}_timeline_overlap.txt")):
    if os.path.exists(os.path.join(G, src)):
        shutil.copy(os.path.join(G, src), os.path.join(P, dst))
write_counters()
print(open(os.path.join(P, f"{tag}_kernel_stats.md")).read())
print({k: (round(v["fetch_bytes"] / 1e6), round(v["write_bytes"] / 1e6)) for k, v in tr.items() if k != "_how"})
