"""Merge the per-file reports of tools/run_reference_tests.sh (and of pytest-xdist workers) into one JSON with a summary."""
import glob, json, os, sys

src, out = sys.argv[1], sys.argv[2]
tests, meta = {}, {}
for path in sorted(glob.glob(os.path.join(src, "reference_tests_part_*.json"))):
    d = json.load(open(path))
    meta = {k: d[k] for k in ("what", "gpu", "have_ase") if k in d}
    tests.update(d["tests"])
summary = {}
for r in tests.values():
    summary[r["class"]] = summary.get(r["class"], 0) + 1
json.dump(dict(meta, summary=summary, tests=tests), open(out, "w"), indent=1)
print(json.dumps(summary))
