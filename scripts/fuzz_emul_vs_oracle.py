"""Randomized parity runs: kernel logic (1-lane emulation, tests/emul) vs the CPU oracle over random run parameters
(w, a, k ranges, filter frequencies, -d, -m, -f, -l, -e), error profiles and coverages.
usage: python scripts/fuzz_emul_vs_oracle.py <seed> <rounds> [--wide] [--w128] [--lanes64] [--warp]      (found the -f / empty pile and the scratch overflow bugs)"""
import sys, os, time, random, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from daccord_amd.synth import SynthData
from daccord_amd._structs import default_params
import pyoracle, emul_lib
seed0=int(sys.argv[1]); nrounds=int(sys.argv[2]); wide = '--wide' in sys.argv; lanes = 64 if '--lanes64' in sys.argv else 1
rng=random.Random(seed0)
bad=0
from common import random_run_config, random_run_config_wide, random_run_config_w128
for r in range(nrounds):
    kw, data, maxin, nplc = (random_run_config_w128 if '--w128' in sys.argv else random_run_config_wide if wide else random_run_config)(rng)
    if '--warp' in sys.argv and not data.get('warp'):   # every round with badly aligned trace blocks
        data['warp'] = (rng.choice([3, 5]), rng.choice([300, 580, 900]), 2000) if data['tspace'] > 125 else (rng.choice([2, 3, 5]), rng.choice([60, 115, 150]))
    try:
        d=SynthData(data['genome_len'],data['nreads'],data['read_len'],**{k:v for k,v in data.items() if k not in ('genome_len','nreads','read_len','profile','warp')})
        prof=data.get('profile') or d.error_profile()
        ovl,piles=pyoracle.pile_select(d.ovl,d.piles,maxinput=maxin)
        npl=min(len(piles),nplc)
        if data.get('warp'):
            from common import warp_trace
            d.trace=warp_trace(ovl,piles,d.trace,range(npl),*data['warp'])
        p=default_params(**kw)
        O=pyoracle.Oracle(p); O.set_error_profile(*prof); O.load_db(d.bps,d.boff,d.rlen)
        E=emul_lib.Emul(p,lanes=lanes); E.set_error_profile(*prof); E.load_db(d.bps,d.boff,d.rlen)
        fo,bo=O.run(piles[:npl],ovl,d.trace,trace_bytes=d.trace_bytes,want_windows=True,nthreads=4); wo=O.windows()
        fe,be=E.run(piles[:npl],ovl,d.trace,trace_bytes=d.trace_bytes); we=E.windows()
        nb=0
        for x,y in zip(wo,we):
            same = x['status']==y['status'] and x['mao']==y['mao'] and x['elength']==y['elength'] and (x['status']!=1 or (bytes(x['cons'])==bytes(y['cons']) and x['minrate']==y['minrate'] and x['filterfreq']==y['filterfreq'] and x['k']==y['k']))
            nb += (not same)
        ok = nb==0 and len(fo)==len(fe) and bo==be and len(wo)==len(we)
        print(("OK  " if ok else "BAD "), r, kw, data, "maxinput",maxin,"piles",npl,"nwin",len(wo),"counts",E.counts(),"wdiff",nb, flush=True)
        bad += (not ok)
    except Exception as ex:
        print("EXC ", r, kw, data, maxin, repr(ex)[:200], flush=True)
        bad += 1
print("DONE bad=%d"%bad, flush=True)
sys.exit(1 if bad else 0)
