"""HBM bytes per launch of the solve kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs).
usage: pmc_traffic.py <dir with pf/ and pw/> <kernel name substring> <out.json> <algorithmic bytes per launch> <workload text>"""
import sys, glob, csv, json, os
o, kname, out, alg, workload = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]


def per_launch(d, name):
    v = []
    for f in glob.glob(os.path.join(o, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kname in r["Kernel_Name"] and r["Counter_Name"] == name:
                v.append(float(r["Counter_Value"]))
    return v


f, w = per_launch("pf", "FETCH_SIZE"), per_launch("pw", "WRITE_SIZE")
fm, wm = sorted(f)[len(f) // 2], sorted(w)[len(w) // 2]
json.dump({"kernel": kname, "workload": workload,
           "collection": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two separate passes (no trace domains), per launch; median over the launches seen",
           "launches_seen": [len(f), len(w)],
           "FETCH_SIZE_KB_per_launch": f if len(f) <= 16 else {"median": fm, "min": min(f), "max": max(f)},
           "WRITE_SIZE_KB_per_launch": w if len(w) <= 16 else {"median": wm, "min": min(w), "max": max(w)},
           "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports half of a coalesced stream's bytes -> doubled "
                         "(the loads here are 8 B/lane, for which the guide gives no calibration: upper estimate); WRITE_SIZE as is",
           "traffic_bytes_per_launch": int((2 * fm + wm) * 1024), "algorithmic_bytes_per_launch": alg},
          open(out, "w"), indent=1)
print(open(out).read()[-300:])
