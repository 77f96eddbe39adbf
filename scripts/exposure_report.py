"""How much of the output depends on the choices the oracle had to make where libmaus2 defines the behaviour (VERDICT r02
task 3)?  The reference cannot be built here, so its co-optimal traceback choice, its heap tie order,
binomRowUpperLimit and its FFT convolution are not known bit for bit.  This script runs the ORACLE (CPU, build container)
on a slice of BASELINE config 2 once per switch (oracle/o_align.hpp: ORACLE_TB_BLOCK, ORACLE_TB_CONS, ORACLE_HEAP_TIE,
ORACLE_KLIM_DELTA, ORACLE_CONV) and reports against the default:
  windows whose record (status, consensus, minrate, filter frequency) changes, corrected bases that change (edit distance
  between the two outputs of every read, fragment lists concatenated), and the checkconsensus error rate against the truth.
usage: python scripts/exposure_report.py [npiles=200] [first=3000] [nthreads=6] [case=cfg2]      -> profiles/r03_exposure[_case].json + table
(one child process per variant: the switches are read when the oracle library is loaded)"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
VARIANTS = [("default", {}),
            ("traceback of the block alignments: diag>ins>del", {"ORACLE_TB_BLOCK": "1"}),
            ("traceback of the block alignments: del>diag>ins", {"ORACLE_TB_BLOCK": "2"}),
            ("traceback of the block alignments: ins>diag>del", {"ORACLE_TB_BLOCK": "3"}),
            ("traceback consensus->A: diag>ins>del", {"ORACLE_TB_CONS": "1"}),
            ("traceback consensus->A: del>diag>ins", {"ORACLE_TB_CONS": "2"}),
            ("traceback consensus->A: ins>diag>del", {"ORACLE_TB_CONS": "3"}),
            ("heap: equal keys rise on push", {"ORACLE_HEAP_TIE": "1"}),
            ("heap: right child among equal children on pop", {"ORACLE_HEAP_TIE": "2"}),
            ("heap: both", {"ORACLE_HEAP_TIE": "3"}),
            ("binomRowUpperLimit - 1", {"ORACLE_KLIM_DELTA": "-1"}),
            ("binomRowUpperLimit + 1", {"ORACLE_KLIM_DELTA": "1"}),
            ("convolution accumulated in long double", {"ORACLE_CONV": "1"}),
            ("convolution summed in descending index order", {"ORACLE_CONV": "2"})]


def child(npiles, first, nthreads, out, cname="cfg2"):
    import numpy as np
    import pyoracle
    from daccord_amd._structs import default_params
    from daccord_amd import checkconsensus
    from scale_cases import CASES, make_case
    case = dict(CASES[cname]); case["first"] = first; case["npiles"] = npiles
    d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
    O = pyoracle.Oracle(default_params(k=14)); O.set_error_profile(*d.error_profile()); O.load_db(d.bps, d.boff, d.rlen)
    t0 = time.time()
    fr, ba = O.run(sel, ovl, d.trace, nthreads=nthreads, want_windows=True)
    w = O.windows()
    _, acc = checkconsensus.check(fr, ba, d.genome, d.truth, d.rlen)
    np.savez(out, status=w["status"], ff=w["filterfreq"], minrate=w["minrate"], cons=np.array([bytes(x) for x in w["cons"]]), elength=w["elength"],
             fr_aread=fr["aread"], fr_first=fr["first"], fr_last=fr["last"], fr_len=fr["len"], fr_off=fr["seq_off"], bases=np.frombuffer(ba, dtype=np.uint8),
             acc=json.dumps(acc), seconds=time.time() - t0)


def main():
    import numpy as np
    npiles = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    nthreads = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    cname = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
    tmp = os.path.join(ROOT, "gpurun_out", "exposure" + ("" if cname == "cfg2" else "_" + cname)); os.makedirs(tmp, exist_ok=True)
    # a private build of the oracle (the switches are compiled in; other test runs may have the shared copy loaded)
    so = os.path.join(tmp, "liboracle_sw.so")
    subprocess.check_call(["g++", "-O3", "-std=c++17", "-fPIC", "-fopenmp", "-ffp-contract=off", "-shared", "-o", so, os.path.join(ROOT, "oracle", "oracle_capi.cpp")])
    os.environ["ORACLE_LIB"] = so
    res = []
    for i, (name, env) in enumerate(VARIANTS):
        out = os.path.join(tmp, "v%02d.npz" % i)
        if not os.path.exists(out):
            e = dict(os.environ); e.update(env)
            subprocess.check_call([sys.executable, __file__, "--child", str(npiles), str(first), str(nthreads), out, cname], env=e)
        res.append((name, env, np.load(out, allow_pickle=False)))
        print("ran", name, "%.0f s" % float(res[-1][2]["seconds"]), flush=True)
    import pyoracle
    L = pyoracle.lib()
    import ctypes as C
    L.oracle_edit_distance.restype = C.c_uint64
    L.oracle_edit_distance.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64]

    def per_read(z):
        by = {}
        b = z["bases"].tobytes()
        for a, o, l in zip(z["fr_aread"], z["fr_off"], z["fr_len"]):
            by.setdefault(int(a), []).append(b[int(o):int(o) + int(l)])
        return {a: b"".join(v) for a, v in by.items()}
    base = res[0][2]; breads = per_read(base)
    nb = sum(len(v) for v in breads.values())
    rows = []
    for name, env, z in res:
        wd = int(((z["status"] != base["status"]) | (z["cons"] != base["cons"]) | (z["minrate"] != base["minrate"]) | (z["ff"] != base["ff"])).sum())
        wc = int((z["cons"] != base["cons"]).sum())
        reads = per_read(z)
        ed = 0; rd = 0
        for a in set(breads) | set(reads):
            x, y = breads.get(a, b""), reads.get(a, b"")
            if x != y:
                rd += 1
                ed += int(L.oracle_edit_distance(x, len(x), y, len(y)))
        acc = json.loads(str(z["acc"]))
        rows.append({"variant": name, "env": env, "windows": int(len(z["status"])), "windows_changed": wd, "windows_consensus_changed": wc,
                     "reads": len(breads), "reads_changed": rd, "corrected_bases": nb, "bases_changed_edit_distance": ed,
                     "frac_windows_changed": round(wd / max(1, len(z["status"])), 6), "frac_bases_changed": round(ed / max(1, nb), 8),
                     "truth_erate": acc.get("erate"), "truth_covered_frac": acc.get("covered_frac")})
    out = {"workload": "tests/scale_cases.py case %s, piles %d..%d, k=14" % (cname, first, first + npiles - 1), "rows": rows}
    json.dump(out, open(os.path.join(ROOT, "profiles", "r03_exposure%s.json" % ("" if cname == "cfg2" else "_" + cname)), "w"), indent=1)
    print("%-52s %10s %10s %12s %12s %10s" % ("variant", "win chg", "cons chg", "bases chg", "frac bases", "erate"))
    for r in rows:
        print("%-52s %10d %10d %12d %12.2e %10.6f" % (r["variant"], r["windows_changed"], r["windows_consensus_changed"], r["bases_changed_edit_distance"], r["frac_bases_changed"], r["truth_erate"] or 0))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], sys.argv[6] if len(sys.argv) > 6 else "cfg2")
    else:
        main()
