"""Per-wave-per-iteration counter values of the fast kernel from the csv files tools/pmc_collect.sh wrote."""
import sys, os, glob, csv, json, collections
out = sys.argv[1]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "mmpc_fast_kernel" not in row.get("Kernel_Name", ""):
            continue
        acc[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]] += 1
iters, waves = 19.0, 1024.0          # probe_one.py "same": instance 0 of the batch, 19 iterations, 1024 one-wave workgroups
res = {k: acc[k] / n[k] / waves / iters for k in sorted(acc)}
print(json.dumps({"note": "rocprofv3 --pmc, tools/probe_one.py 1024 same (1024 identical instances, 19 iterations each): per wave per "
                          "iteration, averaged over the launches seen; SQ_*_CYCLES / WAIT / ACTIVE count quad-cycles",
                  "launches": dict(n), "per_wave_per_iteration": res}, indent=1))
