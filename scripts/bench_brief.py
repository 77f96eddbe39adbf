"""One bench.py JSON line on stdin -> the few numbers a GPU call's tail should show."""
import sys, json
try:
    r = json.loads(sys.stdin.read()); ro = r["roofline"]; pa = r.get("parity", {})
    print("value", r["value"], "e2e", r.get("value_end_to_end"), "incl_plan_h2d", r.get("value_incl_plan_h2d"), "ms", r["ms_per_step"], "kernel_ms", ro["kernel_ms"])
    print("  windows_by_kernel", ro.get("windows_by_kernel"), "handed_on", ro["windows_handed_on"], ro["size_classes"], "identical", pa.get("identical"), pa.get("piles_compared_distinct"), "live", pa.get("live"), pa.get("gpu_fasta_sha256_all", "")[:16])
    print("  roofline", ro["kernel"], ro["achieved"], ro["frac"], ro.get("traffic"), str(ro.get("pmc_source"))[:60])
    cb = r.get("cpu_baseline")
    if cb:
        print("  cpu", cb.get("kind"), cb["value"], cb["cores"], cb.get("identical_to_gpu_on_sample"), (cb.get("port") or {}).get("value"), (cb.get("like_for_like") or {}).get("value"))
except Exception as e:
    print("no json", e)
