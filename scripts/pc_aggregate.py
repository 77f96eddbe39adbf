"""Aggregate a rocprofv3 PC-sampling run on the GPU box (the raw sample table is hundreds of MB and stays there).

usage: pc_aggregate.py <dir with the rocprofv3 csv files> <out.json> [kernel regex = k_window_fast]

Reads  *_pc_sampling_{stochastic,host_trap}.csv  (one row per sampled wavefront: Instruction, Instruction_Comment = file:line when
the library was built with -gline-tables-only, Exec_Mask, Dispatch_Id / Correlation_Id and -- stochastic only --
Wave_Issued_Instruction, Instruction_Type, Stall_Reason, Wave_Count) and  *_kernel_trace.csv  (Dispatch_Id -> kernel name), and
writes per kernel: samples, samples by stall reason / instruction type / issued, by source line, by (line, instruction), the active-lane
histogram, all cut to the top entries.  Column names are looked up in the header, whatever subset this ROCm writes."""
import collections
import csv
import glob
import json
import os
import re
import sys


def find(dirn, pat):
    out = []
    for root, _, files in os.walk(dirn):
        for f in files:
            if re.search(pat, f):
                out.append(os.path.join(root, f))
    return sorted(out)


def main():
    dirn, outfn = sys.argv[1], sys.argv[2]
    kre = re.compile(sys.argv[3] if len(sys.argv) > 3 else "k_window_fast")
    csv.field_size_limit(1 << 30)
    disp = {}
    for fn in find(dirn, r"kernel_trace\.csv$"):
        with open(fn, newline="") as f:
            for r in csv.DictReader(f):
                for key in ("Dispatch_Id", "Correlation_Id"):
                    if key in r:
                        disp[(key, r[key])] = r.get("Kernel_Name", "?")
    res = {"files": [], "kernels": {}}
    for fn in find(dirn, r"pc_sampling.*\.csv$"):
        with open(fn, newline="") as f:
            rd = csv.DictReader(f)
            cols = rd.fieldnames or []
            res["files"].append({"file": os.path.basename(fn), "columns": cols, "size": os.path.getsize(fn)})
            per = {}
            n = 0
            head = []
            for r in rd:
                n += 1
                if n <= 40:
                    head.append(r)
                kn = None
                for key in ("Dispatch_Id", "Correlation_Id"):
                    if key in r and (key, r[key]) in disp:
                        kn = disp[(key, r[key])]
                        break
                kn = kn or "?"
                kn = re.sub(r"^void ", "", kn); kn = re.sub(r"\(.*$", "", kn)
                a = per.get(kn)
                if a is None:
                    a = per[kn] = {"samples": 0, "stall": collections.Counter(), "itype": collections.Counter(), "issued": collections.Counter(),
                                   "line": collections.Counter(), "line_stalled": collections.Counter(), "ins": collections.Counter(), "ins_stall": collections.Counter(), "lanes": collections.Counter(),
                                   "wave_count": collections.Counter()}
                a["samples"] += 1
                ins = r.get("Instruction", "")
                cm = r.get("Instruction_Comment", "")
                cm = re.sub(r"^.*/csrc/", "", cm)
                issued = r.get("Wave_Issued_Instruction", "")
                st = r.get("Stall_Reason", "")
                a["issued"][issued] += 1
                a["stall"][st] += 1
                a["itype"][r.get("Instruction_Type", "")] += 1
                a["line"][cm] += 1
                if issued in ("0", "false", "False"):
                    a["line_stalled"][cm] += 1
                a["ins"][cm + " | " + ins] += 1
                a["ins_stall"][cm + " | " + ins + " | " + st + " | issued=" + issued] += 1
                em = r.get("Exec_Mask", "")
                try:
                    a["lanes"][bin(int(em, 0) if em.startswith("0x") else int(em)).count("1") // 8 * 8] += 1
                except Exception:
                    pass
                wc = r.get("Wave_Count", "")
                if wc:
                    a["wave_count"][wc] += 1
            res["files"][-1]["rows"] = n
            res["files"][-1]["head"] = head
            for kn, a in per.items():
                keep = kre.search(kn) is not None
                top = 400 if keep else 15
                res["kernels"][os.path.basename(fn) + "::" + kn] = {
                    "samples": a["samples"], "stall": dict(a["stall"].most_common(40)), "itype": dict(a["itype"].most_common(40)),
                    "issued": dict(a["issued"]), "lanes_hist_by_8": {str(k): v for k, v in sorted(a["lanes"].items())},
                    "wave_count": dict(a["wave_count"].most_common(40)),
                    "by_line": a["line"].most_common(top), "by_line_not_issued": a["line_stalled"].most_common(top),
                    "by_instruction": a["ins"].most_common(top * 2), "by_instruction_stall": a["ins_stall"].most_common(top * 2)}
    with open(outfn, "w") as f:
        json.dump(res, f, indent=0)
    for k, v in res["kernels"].items():
        print(k, v["samples"], list(v["stall"].items())[:6])


if __name__ == "__main__":
    main()
