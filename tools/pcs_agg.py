"""Aggregate rocprofv3 PC-sampling csv output: samples per code-object offset (and per instruction text if present)."""
import sys, glob, csv, collections, os
d = sys.argv[1]
files = [f for f in glob.glob(os.path.join(d, "**", "*.csv"), recursive=True)]
print("files:", [(f, os.path.getsize(f)) for f in files])
for f in files:
    if "pc_sampling" not in f: continue
    with open(f) as fh:
        r = csv.reader(fh); hdr = next(r); print("HEADER", hdr)
        cols = {h: i for i, h in enumerate(hdr)}
        key = [i for h, i in cols.items() if "offset" in h.lower() or h.lower() in ("pc", "instruction")]
        C = collections.Counter(); n = 0; first = []
        for row in r:
            if n < 3: first.append(row)
            n += 1
            C[tuple(row[i] for i in key)] += 1
        print("rows", n, "first", first)
        with open(os.path.join(d, "pcs_summary.txt"), "a") as o:
            o.write("# %s rows %d key %s\n" % (f, n, [hdr[i] for i in key]))
            for k, v in sorted(C.items(), key=lambda kv: -kv[1])[:4000]:
                o.write("%d\t%s\n" % (v, "\t".join(k)))
