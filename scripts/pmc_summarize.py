"""Summarise the rocprofv3 --pmc passes (scripts/gpu_pmc.sh) into profiles/<tag>_pmc_summary.json.
The summary records the hash of daccord_amd/csrc it was collected on; bench.py quotes it only while that hash matches.

Per kernel (mean over its launches in the pass):
  traffic            2*FETCH_SIZE + WRITE_SIZE (KB -> bytes; FETCH_SIZE is doubled on gfx950, MI355X_MICROARCH.md HBM section;
                     separate passes)
  issue fractions    per resident wave: SQ_INSTS_VALU / SQ_WAVE_CYCLES (both in units of four clocks), ...
  resident waves     per CU = 4*SQ_WAVE_CYCLES / (kernel duration in shader cycles * 256 CUs)
  diagnostics (round 4, collected on a smaller slice of the same workload; ratios):
    valu_lane_util         SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU): mean share of the 64 lanes a VALU instruction has active
    inflight_lds/vmem/smem SQ_INST_LEVEL_* = sum over cycles of the instructions in flight: the time-weighted share of each memory
                           class among what the wavefronts have outstanding (LDS round trips vs slab / table loads vs scalar loads);
                           per instruction: SQ_INST_LEVEL_x / SQ_INSTS_x = mean time in flight (counter units)
    tcc_hit_rate           TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
    tcp_tcc_read_latency   TCP_TCC_READ_REQ_LATENCY_sum / TCP_TCC_READ_REQ_sum (cycles from L1 miss to L2 data)
    lds_bank_conflict      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
    active_valu/scalar/lds SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES"""
import csv, json, os, sys, collections

def load(d):
    fn = os.path.join(d, "pmc_counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    if not os.path.exists(fn):
        return acc, dur
    per = collections.defaultdict(lambda: collections.defaultdict(float)); seen = {}
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"].split("(")[0]
        for a, b in (("void ", ""), ("dacc::", "")):
            k = k.replace(a, b)
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])      # (a counter may be reported per XCD / dimension)
        seen[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    for (k, disp), cs in per.items():
        for c, v in cs.items():
            acc[k][c].append(v)
        dur[k].append(seen[(k, disp)])
    return acc, dur

def main():
    root, reads, readlen, cov, k = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4]), int(sys.argv[5])
    clock_ghz = float(sys.argv[6]) if len(sys.argv) > 6 else 2.4
    tag = sys.argv[7] if len(sys.argv) > 7 else "r04"
    dreads = int(sys.argv[8]) if len(sys.argv) > 8 else reads
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from daccord_amd import build as _b
    out = {"workload": {"reads": reads, "readlen": readlen, "coverage": cov, "k": k}, "diagnostics_workload_reads": dreads, "csrc_hash": _b.csrc_hash(), "kernel_isa": _b.kernel_isa_hashes(), "kernels": {},
           "note": "rocprofv3 --pmc, one pass per counter group, mean per launch; traffic = 2*FETCH_SIZE + WRITE_SIZE (KB->bytes); "
                   "the diagnostic groups (lane utilisation, in-flight levels, L2 hit rate, L1->L2 latency) were collected on a smaller slice: ratios only"}
    F, dF = load(os.path.join(root, "pmc_FETCH_SIZE")); W, dW = load(os.path.join(root, "pmc_WRITE_SIZE")); S, dS = load(os.path.join(root, "pmc_SQ_WAVE_CYCLES"))
    LN, _ = load(os.path.join(root, "pmc_LANES")); AC, _ = load(os.path.join(root, "pmc_ACTIVITY")); TC, _ = load(os.path.join(root, "pmc_TCC")); TP, _ = load(os.path.join(root, "pmc_TCP"))
    mean = lambda v: sum(v) / len(v) if v else 0.0
    for kn in sorted(set(F) | set(W) | set(S) | set(LN)):
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
        ln = {c: mean(v) for c, v in LN.get(kn, {}).items()}; ac = {c: mean(v) for c, v in AC.get(kn, {}).items()}
        tc = {c: mean(v) for c, v in TC.get(kn, {}).items()}; tp = {c: mean(v) for c, v in TP.get(kn, {}).items()}
        dg = {}
        if ln.get("SQ_ACTIVE_INST_VALU"):
            dg["valu_lane_util"] = round(ln["SQ_THREAD_CYCLES_VALU"] / (64.0 * ln["SQ_ACTIVE_INST_VALU"]), 4)
            dg["active_valu_frac"] = round(ln["SQ_ACTIVE_INST_VALU"] / ln["SQ_WAVE_CYCLES"], 4)
            lv = {"lds": ln.get("SQ_INST_LEVEL_LDS", 0.0), "vmem": ln.get("SQ_INST_LEVEL_VMEM", 0.0), "smem": ac.get("SQ_INST_LEVEL_SMEM", 0.0)}
            tot = sum(lv.values())
            if tot:
                dg["inflight_share"] = {a: round(b / tot, 4) for a, b in lv.items()}
            nv = ln.get("SQ_INSTS_VMEM_RD", 0.0) + ln.get("SQ_INSTS_VMEM_WR", 0.0)
            dg["mean_inflight_units_per_instruction"] = {"lds": round(lv["lds"] / ln["SQ_INSTS_LDS"], 3) if ln.get("SQ_INSTS_LDS") else None,
                                                         "vmem": round(lv["vmem"] / nv, 3) if nv else None,
                                                         "smem": round(lv["smem"] / ac["SQ_INSTS_SMEM"], 3) if ac.get("SQ_INSTS_SMEM") else None}
            dg["instructions_per_launch"] = {"lds": ln.get("SQ_INSTS_LDS"), "vmem_rd": ln.get("SQ_INSTS_VMEM_RD"), "vmem_wr": ln.get("SQ_INSTS_VMEM_WR")}
        if ac.get("SQ_WAVE_CYCLES"):
            dg["active_scalar_frac"] = round(ac.get("SQ_ACTIVE_INST_SCA", 0.0) / ac["SQ_WAVE_CYCLES"], 4)
            dg["active_lds_frac"] = round(ac.get("SQ_ACTIVE_INST_LDS", 0.0) / ac["SQ_WAVE_CYCLES"], 4)
            if ac.get("SQ_LDS_IDX_ACTIVE"):
                dg["lds_bank_conflict"] = round(ac.get("SQ_LDS_BANK_CONFLICT", 0.0) / ac["SQ_LDS_IDX_ACTIVE"], 4)
        if tc.get("TCC_HIT_sum") is not None and (tc.get("TCC_HIT_sum", 0) + tc.get("TCC_MISS_sum", 0)):
            dg["tcc_hit_rate"] = round(tc["TCC_HIT_sum"] / (tc["TCC_HIT_sum"] + tc["TCC_MISS_sum"]), 4)
        if tp.get("TCP_TCC_READ_REQ_sum"):
            dg["tcp_tcc_read_latency_cycles"] = round(tp["TCP_TCC_READ_REQ_LATENCY_sum"] / tp["TCP_TCC_READ_REQ_sum"], 1)
            dg["tcp_tcc_read_req"] = tp["TCP_TCC_READ_REQ_sum"]; dg["tcp_tcc_write_req"] = tp.get("TCP_TCC_WRITE_REQ_sum")
        if dg:
            e["diagnostics"] = dg
        out["kernels"][kn] = e
    json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "%s_pmc_summary.json" % tag), "w"), indent=1)
    print(json.dumps({k: {a: b for a, b in v.items() if a != "sq"} for k, v in out["kernels"].items()}, indent=1))

main()
