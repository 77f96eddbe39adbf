"""Randomized parity runs: kernel logic (1-lane emulation, tests/emul) vs the CPU oracle over random run parameters
(w, a, k ranges, filter frequencies, -d, -m, -f, -l, -e), error profiles and coverages.
usage: python scripts/fuzz_emul_vs_oracle.py <seed> <rounds>      (found the -f / empty pile and the scratch overflow bugs)"""
import sys, os, time, random, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from daccord_amd.synth import SynthData
from daccord_amd._structs import default_params
import pyoracle, emul_lib
seed0=int(sys.argv[1]); nrounds=int(sys.argv[2])
rng=random.Random(seed0)
bad=0
for r in range(nrounds):
    w=rng.choice([24,32,40,40,40,48,56,63]); a=rng.choice([5,8,10,10,16,20]); a=min(a,w)
    klow=rng.choice([6,7,8,8,9,10,12,14,14,16]); khigh=klow+rng.choice([0,0,0,1,2]); khigh=min(khigh,16)
    if klow>=w-4: klow=khigh=8
    erate=rng.choice([0.02,0.08,0.12,0.15,0.15,0.2,0.28])
    mix=rng.choice([(0.8,0.1333,0.0667),(1/3,1/3,1/3),(0.5,0.4,0.1),(0.2,0.7,0.1)])
    nreads=rng.choice([60,120,200,300]); rlen=rng.choice([1500,3000,5000]); glen=rng.choice([30000,60000,100000])
    kw=dict(w=w,a=a,klow=klow,khigh=khigh)
    if rng.random()<0.3: kw['maxalign']=rng.choice([3,5,8,15])
    if rng.random()<0.2: kw['minwindowcov']=rng.choice([2,4,5])
    if rng.random()<0.2: kw['producefull']=1
    if rng.random()<0.2: kw['minlen']=rng.choice([200,1000])
    if rng.random()<0.2: kw['maxfilterfreq']=rng.choice([1,3]);
    if rng.random()<0.15: kw['minfilterfreq']=1
    if rng.random()<0.1: kw['eminrate']=rng.choice([5,15,40])
    seed=rng.randrange(1,10**6)
    try:
        d=SynthData(glen,nreads,rlen,erate=erate,seed=seed,ins_frac=mix[0],del_frac=mix[1],sub_frac=mix[2])
        ovl,piles=pyoracle.pile_select(d.ovl,d.piles,maxinput=rng.choice([5000,5000,10]))
        npl=min(len(piles),rng.choice([2,3,4]))
        p=default_params(**kw)
        O=pyoracle.Oracle(p); O.set_error_profile(*d.error_profile()); O.load_db(d.bps,d.boff,d.rlen)
        E=emul_lib.Emul(p); E.set_error_profile(*d.error_profile()); E.load_db(d.bps,d.boff,d.rlen)
        fo,bo=O.run(piles[:npl],ovl,d.trace,want_windows=True,nthreads=1); wo=O.windows()
        fe,be=E.run(piles[:npl],ovl,d.trace); we=E.windows()
        nb=0
        for x,y in zip(wo,we):
            same = x['status']==y['status'] and x['mao']==y['mao'] and x['elength']==y['elength'] and (x['status']!=1 or (bytes(x['cons'])==bytes(y['cons']) and x['minrate']==y['minrate'] and x['filterfreq']==y['filterfreq'] and x['k']==y['k']))
            nb += (not same)
        ok = nb==0 and len(fo)==len(fe) and bo==be and len(wo)==len(we)
        print(("OK  " if ok else "BAD "), r, kw, "erate",erate,"mix",mix[0],"seed",seed,"n",nreads,rlen,glen,"piles",npl,"nwin",len(wo),"counts",E.counts(),"wdiff",nb, flush=True)
        bad += (not ok)
    except Exception as ex:
        print("EXC ", r, kw, erate, mix, seed, nreads, rlen, glen, repr(ex)[:200], flush=True)
        bad += 1
print("DONE bad=%d"%bad, flush=True)
sys.exit(1 if bad else 0)
