"""Instruction / wave-time ledger of the window kernels by phase (VERDICT r05 task 1b), on the GPU box.

The -DDACC_LEDGER build of the library (daccord_amd/libvar_ledger.so; fast_window.hpp LEDGER_REP) runs the phase whose bit is set in
DACC_LEDGER_MASK twice; every listed phase is idempotent, so the output does not move (checked: the FASTA digest of every pass must equal
the baseline's) and the difference of the SQ counters of a kernel between the pass with the bit and the pass without is the phase's own
cost: instructions by class and wave time (SQ_WAVE_CYCLES counts quad-cycles: x 4 = shader cycles), per window of that kernel.  What
cannot run twice -- the lane 0 replay of the pairs into the candidate heap, candidate ranking / decoding, addNextFromHeap, hand-over
slots -- is the remainder (total minus the phases).

usage: python scripts/ledger.py <outdir> [reads=1500] [extra bench.py args...]        (needs rocprofv3; ~25 s per pass, 19 passes)
The library: python -c "from daccord_amd import build; build.build_variant('ledger', ['-DDACC_LEDGER'])"   (here, before the GPU call)"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PHASES = [(0, "gather: overlap selection, string descriptors, bases -> pattern masks"), (1, "estimateLength (f64 products over the strings)"),
          (2, "buildInstances: k-mer instances + register bitonic sorts"), (3, "buildNodes: nodes, positions, support ranges, first / last k-mer lists"),
          (16, "passIsDead"), (17, "saveInstances (sorted instances -> slab)"), (4, "buildSuccessors"),
          (5, "computeBaseStretches: predecessor counts, walks, links, two sorts"), (6, "findCandidatesAndPieces"), (7, "pairReachable (exact prune)"),
          (8, "computeStretchFeasLanes: stretch feasibility + weight records"), (9, "spillS (build-phase arrays -> slab)"),
          (10, "reverse enumerations of all last k-mers + block copy / sort / rank"), (11, "forward trees of a batch of first k-mers + finish"),
          (12, "pairs, lane-parallel part: classify, matchPops, interval tasks, heaps + pops"), (13, "restoreS (slab -> build-phase arrays)"),
          (14, "candidate errors: Myers distance of every (candidate, string)"), (15, "alignAndEmit (lane 0) + emitRecord")]
COUNTERS = ["SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"]


def one_pass(out, tag, mask, reads, extra):
    d = os.path.join(out, "pass_" + tag)
    env = dict(os.environ, DACC_LIB=os.path.join(ROOT, "daccord_amd", "libvar_ledger.so"), DACC_LEDGER_MASK=str(mask), TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "pmc", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--reads", str(reads), "--steps", "1", "--warmup", "0", "--no-cpu", "--e2e-steps", "0"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd="/tmp", timeout=600)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if r.returncode != 0 or not line:
        raise RuntimeError("pass %s failed: %s" % (tag, (r.stderr or r.stdout)[-1500:]))
    res = json.loads(line[-1])
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0].replace("void ", "")][row["Counter_Name"]] += float(row["Counter_Value"])
    subprocess.run(["rm", "-rf", d])
    return res, {k: dict(v) for k, v in acc.items()}


def main():
    out = sys.argv[1]; reads = int(sys.argv[2]) if len(sys.argv) > 2 else 1500; extra = sys.argv[3:]
    os.makedirs(out, exist_ok=True)
    base_res, base = one_pass(out, "base", 0, reads, extra)
    sha = base_res["parity"]["gpu_fasta_sha256_all"]
    wins = base_res["roofline"]["windows_by_kernel"]; launches = 2      # first pass + one step
    kern = {"k_window_fast<0>": wins.get("k_classify+k_window_fast<0>", 0), "k_window_fast<7>": wins.get("k_window_fast<7>", 0), "k_window_fast<1>": wins.get("k_window_fast<1>", 0),
            "k_window_fast<6>": wins.get("k_window_fast<6>", 0)}
    rows = {}
    for bit, name in PHASES:
        res, acc = one_pass(out, "p%02d" % bit, 1 << bit, reads, extra)
        same = res["parity"]["gpu_fasta_sha256_all"] == sha
        rows[bit] = {"name": name, "identical": same, "delta": {k: {c: acc.get(k, {}).get(c, 0.0) - base.get(k, {}).get(c, 0.0) for c in COUNTERS} for k in kern}}
        print("phase %2d %-70s identical=%s" % (bit, name[:70], same), flush=True)
    json.dump({"reads": reads, "extra": extra, "windows": kern, "launches": launches, "baseline": {k: base.get(k, {}) for k in kern}, "phases": rows, "fasta_sha256": sha},
              open(os.path.join(out, "ledger.json"), "w"), indent=1)
    for k, nw in kern.items():
        if not nw:
            continue
        per = 1.0 / (nw * launches)
        b = base[k]
        tot_i = sum(b[c] for c in COUNTERS[2:])
        print("\n== %s: %d windows per launch; per window: %.0f instructions (VALU %.0f, SALU %.0f, branch %.0f, LDS %.0f, VMEM %.0f), %.0f wave cycles (%.0f waiting)"
              % (k, nw, tot_i * per, b["SQ_INSTS_VALU"] * per, b["SQ_INSTS_SALU"] * per, b["SQ_INSTS_BRANCH"] * per, b["SQ_INSTS_LDS"] * per,
                 (b["SQ_INSTS_VMEM_RD"] + b["SQ_INSTS_VMEM_WR"]) * per, 4 * b["SQ_WAVE_CYCLES"] * per, 4 * b["SQ_WAIT_ANY"] * per))
        print("%-72s %8s %7s %7s %7s %6s %6s %9s %9s %6s %6s" % ("phase (per window of this kernel)", "instr", "VALU", "SALU", "branch", "LDS", "VMEM", "cycles", "waiting", "%ins", "%cyc"))
        si = sc = 0.0
        for bit, name in PHASES:
            dl = rows[bit]["delta"][k]
            ins = sum(dl[c] for c in COUNTERS[2:]) * per; cyc = 4 * dl["SQ_WAVE_CYCLES"] * per
            si += ins; sc += cyc
            print("%-72s %8.0f %7.0f %7.0f %7.0f %6.0f %6.0f %9.0f %9.0f %6.1f %6.1f" % (name[:72], ins, dl["SQ_INSTS_VALU"] * per, dl["SQ_INSTS_SALU"] * per, dl["SQ_INSTS_BRANCH"] * per,
                  dl["SQ_INSTS_LDS"] * per, (dl["SQ_INSTS_VMEM_RD"] + dl["SQ_INSTS_VMEM_WR"]) * per, cyc, 4 * dl["SQ_WAIT_ANY"] * per, 100 * ins / (tot_i * per), 100 * cyc / (4 * b["SQ_WAVE_CYCLES"] * per)))
        print("%-72s %8.0f %58s %9.0f %16.1f %6.1f" % ("remainder (lane 0 replay of the pairs, ranking / decode, addNextFromHeap, ...)", tot_i * per - si, "", 4 * b["SQ_WAVE_CYCLES"] * per - sc,
              100 - 100 * si / (tot_i * per), 100 - 100 * sc / (4 * b["SQ_WAVE_CYCLES"] * per)))


if __name__ == "__main__":
    main()
