"""Files the two bench lines of tools/bench_both.sh (gpurun_out/bench_full.json, bench_window.json) as profiles/<tag>_bench*.json."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "round4"
for f, o in (("bench_full.json", f"{tag}_bench.json"), ("bench_window.json", f"{tag}_bench_driver_window.json")):
    ln = [l for l in open(os.path.join(ROOT, "gpurun_out", f)).read().strip().splitlines() if l.startswith("{")][-1]
    d = json.loads(ln)
    json.dump(d, open(os.path.join(ROOT, "profiles", o), "w"), indent=1)
    print(o, "value %.0f" % d["value"], "outs_on %.0f" % d["outs_on"]["value"], "config5 %.0f" % d["config5"]["default"]["value"],
          "dense %.0f" % d["kernels"]["ql0001_dense"]["value"], "frac %.4f" % d["roofline"]["frac"])
