"""Prints the parts of a bench.py JSON line that change between rounds (headline, frontends, loi block)."""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print("headline: %.4g %s, %.4f ms/step, frontend %s, kernel %.4f ms, frac %.3f, traffic %s, pmc_error %s" % (
    d["value"], d["unit"], d["ms_per_step"], d["config"]["frontend"], r["kernel_ms"], r.get("frac", float("nan")), r.get("traffic"), r.get("pmc_error")))
print("frontends:", json.dumps(d.get("frontends")))
for e in d.get("sweep", []):
    print("sweep %-16s D=%3d %s" % (e["workload"], e["dim"], ("%.4f ms  dense %d  frac %.3f" % (e["kernel_ms"], e["dense_windows"], e["roofline"].get("frac", float("nan")))) if "error" not in e else e["error"]))
print("fused:", json.dumps(d.get("fused")))
l = d.get("loi", {})
print(json.dumps({k: v for k, v in l.items() if k != "cases"}, indent=1))
for c in l.get("cases", []):
    print("%-18s D=%3d rule %d: %.4f ms  dense windows %6d (%.1f %%)  fabric %.3f GB  L2 hit %.3f  MFMA busy %.2f %%  oracle %s" % (
        c["graph"], c["dim"], c["rule"], c["kernel_ms"], c["dense_windows"], 100 * c["dense_window_share"], c.get("fabric_bytes", 0) / 1e9,
        c.get("l2_hit_rate", 0), c.get("mfma_busy_percent", 0), c["oracle_check"]))
print("cpu_baseline:", json.dumps({k: v for k, v in d.get("cpu_baseline", {}).items() if k != "oracle_port"}))
