"""CPU design aid: run the 1-lane host emulation of the window kernels on a slice of BASELINE config 2 and report how many
windows each capacity tier hands on and why (DACC_EMUL_OVER lines).  No GPU needed; the counts are deterministic."""
import os, sys, re, subprocess, collections, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
if os.environ.get("DACC_EMUL_OVER") is None and "--child" not in sys.argv:
    env = dict(os.environ, DACC_EMUL_OVER="1")
    p = subprocess.run([sys.executable, __file__, "--child"] + sys.argv[1:], env=env, stderr=subprocess.PIPE, text=True)
    cnt = collections.Counter()
    for l in p.stderr.splitlines():
        m = re.match(r"\[over\] tier maxs=(\d+) line (\d+) bits (0x[0-9a-f]+)", l)
        if m:
            cnt[(int(m.group(1)), m.group(3), int(m.group(2)))] += 1
        elif l.strip():
            print(l)
    for key, v in sorted(cnt.items()):
        print("tier maxs=%d bits=%s line=%d : %d" % (key[0], key[1], key[2], v))
    sys.exit(p.returncode)
import numpy as np
import emul_lib
from daccord_amd import engine  # noqa (host-only pile_select lives in the io lib)
from daccord_amd._structs import default_params
from scale_cases import CASES, make_case
import pyoracle
args = [a for a in sys.argv[1:] if a != "--child"]
npiles = int(args[0]) if args else 16
first = int(args[1]) if len(args) > 1 else 0
k = int(args[2]) if len(args) > 2 else 14
cname = args[3] if len(args) > 3 else "cfg2"
case = dict(CASES[cname]); case["first"] = first; case["npiles"] = npiles
d, ovl, piles, sel = make_case(case, pyoracle.pile_select)
pkw = dict(k=k)
if os.environ.get("STAT_W"):      # wide windows: STAT_W=80 (advance w/4 unless STAT_A)
    pkw["w"] = int(os.environ["STAT_W"]); pkw["a"] = int(os.environ.get("STAT_A", str(max(1, pkw["w"] // 4))))
E = emul_lib.Emul(default_params(**pkw)); E.set_error_profile(*d.error_profile()); E.load_db(d.bps, d.boff, d.rlen)
t0 = time.time()
fr, ba = E.run(sel, ovl, d.trace)
w = E.windows()
print("piles=%d windows=%d tiers(t1,t2,t3,generic)=%s tier0/7/10=%s  %.1fs  ff=%s" % (npiles, len(w), E.counts(), (E.count_tier0(), E.count_tier7(), E.count_tier10()), time.time() - t0,
      dict(zip(*np.unique(w["filterfreq"], return_counts=True)))), file=sys.stderr)
