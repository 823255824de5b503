"""one line per bench.py JSON line: rate, per-kernel launch times, the chained launch's device-clock percentiles, CPU ratio"""
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
except Exception as e:
    print("no bench line in", sys.argv[1], e); sys.exit(0)
k = d["roofline"].get("kernels")
if not k or len(k) < 4:      # (the informational --chains line: no per-kernel block)
    print("%-34s %9d  %s" % ((sys.argv[2] if len(sys.argv) > 2 else "")[:34], round(d["value"]), json.dumps(d["roofline"])[:160])); sys.exit(0)
s = "%-34s %9d  " % ((sys.argv[2] if len(sys.argv) > 2 else "")[:34], round(d["value"]))
s += "A %.2f + %.2f us  P %.2f + %.2f us  " % (k[0]["avg_launch_us"], k[2]["avg_launch_us"], k[1]["avg_launch_us"], k[3]["avg_launch_us"])
for kk in k[:2]:
    if "launch_us_percentiles" in kk:
        p = kk["launch_us_percentiles"]
        s += "[%s inside %.2f p50 %.1f p75 %.1f p90 %.1f p99 %.1f; events %.2f] " % (kk["sampler"], kk.get("inside_launch_us", 0.0), p["p50_us"], p["p75_us"], p["p90_us"], p["p99_us"], kk["avg_launch_us_hip_event_sample"])
s += "frac %s kt/wall %.3f" % (("%.4f" % d["roofline"]["frac"]) if d["roofline"]["frac"] else "None", d["roofline"].get("kernel_time_over_wall") or 0.0)
cb = d.get("cpu_baseline")
if cb and cb.get("value"):
    s += "  cpu %.3g (%s thr) x%.2f  by_threads %s" % (cb["value"], cb["cores"], d["value"] / cb["value"], [(b["threads"], round(b["value"])) for b in cb["by_threads"]])
print(s)
