"""Summarise the rocprofv3 --pmc passes (scripts/gpu_pmc.sh) into profiles/<tag>_pmc_summary.json (tag: last argument, default r03).
The summary records the hash of daccord_amd/csrc it was collected on; bench.py quotes it only while that hash matches.

Per kernel (mean over its launches in the pass): HBM traffic = 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE is
doubled on gfx950, MI355X_MICROARCH.md HBM section; separate passes), issue fractions from the SQ counters
(per resident wave: SQ_INSTS_VALU / SQ_WAVE_CYCLES, both in units of four clocks), resident waves per
CU = 4*SQ_WAVE_CYCLES / (kernel duration in shader cycles * 256 CUs)."""
import csv, json, os, sys, collections

def load(d):
    fn = os.path.join(d, "pmc_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    if not os.path.exists(fn):
        return acc, dur
    seen = set()
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        for a, b in (("void ", ""), ("dacc::", "")):
            k = k.replace(a, b)
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6)
    return acc, dur

def main():
    root, reads, readlen, cov, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    clock_ghz = float(sys.argv[6]) if len(sys.argv) > 6 else 2.4
    tag = sys.argv[7] if len(sys.argv) > 7 else "r03"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from daccord_amd import build as _b
    out = {"workload": {"reads": reads, "readlen": readlen, "coverage": cov, "k": k}, "csrc_hash": _b.csrc_hash(), "kernels": {},
           "note": "rocprofv3 --pmc, one pass per counter group (FETCH_SIZE | WRITE_SIZE | SQ_*), mean per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KB->bytes)"}
    F, dF = load(os.path.join(root, "pmc_FETCH_SIZE")); W, dW = load(os.path.join(root, "pmc_WRITE_SIZE")); S, dS = load(os.path.join(root, "pmc_SQ_WAVE_CYCLES"))
    mean = lambda v: sum(v) / len(v) if v else 0.0
    for kn in sorted(set(F) | set(W) | set(S)):
        if not kn.startswith("k_"):
            continue
        e = {"launches": len(dF.get(kn, [])) or len(dS.get(kn, []))}
        f = mean(F[kn].get("FETCH_SIZE", [])); w = mean(W[kn].get("WRITE_SIZE", []))
        e["fetch_kb"] = f; e["write_kb"] = w; e["traffic_bytes_per_launch"] = (2 * f + w) * 1024.0
        e["pmc_kernel_ms"] = round(mean(dF.get(kn, []) or dS.get(kn, [])), 3)
        s = S.get(kn, {})
        wc = mean(s.get("SQ_WAVE_CYCLES", []))
        if wc:
            ms = mean(dS.get(kn, []))
            # SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (4 clocks); a wave64 VALU or LDS instruction holds its SIMD's
            # issue port for one quad-cycle, a SALU instruction for one clock
            e["valu_issue_frac"] = round(mean(s.get("SQ_INSTS_VALU", [])) / wc, 4)
            e["salu_issue_frac"] = round(mean(s.get("SQ_INSTS_SALU", [])) / (4.0 * wc), 4)
            e["lds_issue_frac"] = round(mean(s.get("SQ_INSTS_LDS", [])) / wc, 4)
            e["wait_frac"] = round(mean(s.get("SQ_WAIT_ANY", [])) / wc, 4) if s.get("SQ_WAIT_ANY") else None
            e["resident_waves_per_cu"] = round(4.0 * wc / (ms * 1e-3 * clock_ghz * 1e9 * 256), 2) if ms else None
            e["sq"] = {c: mean(v) for c, v in s.items()}
        out["kernels"][kn] = e
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "sq"} for k, v in out["kernels"].items()}, indent=1))

main()
