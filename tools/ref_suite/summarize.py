"""Per-test-function pass / fail / skip table of a tools/ref_suite/run.sh run (gpurun_out/ref_suite/*.jsonl)."""
import glob
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ref_suite"
tot = {"passed": 0, "failed": 0, "skipped": 0}
fails = []
print(f"{'suite':6s} {'test function':44s} {'parametrised':>12s} {'sampled':>8s} {'passed':>7s} {'failed':>7s} {'skipped':>8s}")
for f in sorted(glob.glob(os.path.join(d, "*.jsonl"))):
    suite = os.path.basename(f).split("_")[0]
    for line in open(f):
        r = json.loads(line)
        for fn in sorted(r["totals"]):
            t, c = r["totals"][fn], r["counts"].get(fn, {"passed": 0, "failed": 0, "skipped": 0})
            print(f"{suite:6s} {fn:44s} {t['parametrised']:12d} {t['sampled']:8d} {c['passed']:7d} {c['failed']:7d} {c['skipped']:8d}")
            for k in tot:
                tot[k] += c[k]
        if r.get("not_applicable"):
            print(f"{suite:6s}   ({r['not_applicable']} cases deselected as not applicable to a ROCm backend: id regex {r['deselect']!r})")
        fails += [(suite, x) for x in r["failures"]]
print(f"TOTAL passed {tot['passed']} failed {tot['failed']} skipped {tot['skipped']}")
for suite, x in fails[:60]:
    print("\nFAIL", suite, x["id"])
    print("   ", x["msg"].strip().splitlines()[-1][:300] if x["msg"].strip() else "")
