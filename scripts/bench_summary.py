"""print the timings of a bench.py JSON line:  python scripts/bench_summary.py gpurun_out/bench.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g %s   ms_per_step %.6f   roofline frac %s" % (d["value"], d["unit"], d["ms_per_step"], d["roofline"].get("frac")))
for k, v in d.items():
    if isinstance(v, dict) and k not in ("config", "roofline", "cpu_baseline", "cpu_baseline_array_formulation"):
        t = {a: round(b, 2) for a, b in v.items() if isinstance(b, (int, float)) and ("us" in a or "seconds" in a)}
        for a, b in v.items():
            if isinstance(b, dict):
                t.update({a + "." + x: round(y, 2) for x, y in b.items() if isinstance(y, (int, float)) and "us" in x})
        print(k, t)
