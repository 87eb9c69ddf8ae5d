"""prints the fields of a bench line that the round's notes quote (python bench.py ... > file)"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
sb = d.get("small_batches", {})
print("value %.4g  frac %.3f  stale %s  cpu %.4g  1 walker %.2f us  64 walkers %.2f us  build %s" % (
    d["value"], d["roofline"]["frac"], d["roofline"].get("instr_source_stale"),
    (d.get("cpu_baseline") or {}).get("value", float("nan")),
    sb.get("walkers_1", {}).get("us_per_iteration", float("nan")),
    sb.get("walkers_64", {}).get("us_per_iteration", float("nan")), d["build"]["id"]))
