#!/usr/bin/env python3
"""Differential fuzz against the imported reference (build container only, no GPU): see the probe table below.
Reads stay inside the streams (the reference's own generators raise RuntimeError at the end of a finite stream on Python >= 3.7)."""
import sys, random, warnings, itertools
warnings.filterwarnings("ignore")
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/reference')
import audiolazy as ref, audiolazy_amd as own
def nv(m,v):
    if isinstance(v,bool) or v is None: return v
    if isinstance(v,int): return ("i",v)
    if isinstance(v,float): return ("f",v.hex())
    if isinstance(v,complex): return ("c",v.real.hex(),v.imag.hex())
    if hasattr(v,'take'): return ("S",[nv(m,x) for x in v.take(3)])
    if isinstance(v,(list,tuple)): return [nv(m,x) for x in v]
    return ("?",type(v).__name__)
def nf(m,f):
    if isinstance(f,(m.CascadeFilter,m.ParallelFilter)): return (type(f).__name__,[nf(m,x) for x in f])
    return ("Z",[(nv(m,k),nv(m,c)) for k,c in f.numpoly.terms()],[(nv(m,k),nv(m,c)) for k,c in f.denpoly.terms()])
def outcome(m,fn):
    try: return nf(m,fn())
    except Exception as e: return ("raises",type(e).__name__)
rng=random.Random(5)
bad={}
N=int(sys.argv[1]) if len(sys.argv)>1 else 200
for i in range(N):
    f=[rng.uniform(.01,3.) for _ in range(5)]; b=[rng.uniform(.001,.5) for _ in range(5)]
    sf=rng.random()<.7; sb=rng.random()<.5
    def F(m): return m.Stream(*f) if sf else f[0]
    def B(m): return m.Stream(*b) if sb else b[0]
    P={}
    for nm in ["poles_exp","freq_poles_exp","z_exp","freq_z_exp"]:
        P["res."+nm]=lambda m,nm=nm: getattr(m.resonator,nm)(F(m),B(m))
    for nm in ["pole","pole_exp","z","z_exp"]:
        P["lp."+nm]=lambda m,nm=nm: getattr(m.lowpass,nm)(F(m))
        P["hp."+nm]=lambda m,nm=nm: getattr(m.highpass,nm)(F(m))
    P["gt.klapuri"]=lambda m: m.gammatone.klapuri(F(m),B(m))
    P["gt.slaney"]=lambda m: m.gammatone.slaney(f[0],b[0])
    P["gt.sampled"]=lambda m: m.gammatone.sampled(f[0],b[0])
    P["gt.sampled_phase"]=lambda m: m.gammatone.sampled(f[0],b[0],phase=.3,eta=3)
    P["comb.fb"]=lambda m: m.comb.fb(rng and 7, F(m)/4 if sf else .5)
    P["comb.tau"]=lambda m: m.comb.tau(5.5, 20)
    P["comb.ff"]=lambda m: m.comb.ff(3, B(m))
    P["alg"]=lambda m: (1 - F(m) * m.z ** -1) / (1 + B(m) * m.z ** -2) * 2
    P["alg2"]=lambda m: (m.resonator(F(m),B(m)) + m.lowpass(f[1])) 
    P["alg3"]=lambda m: m.resonator(f[0],b[0]) * m.lowpass(F(m))
    for name,p in P.items():
        x=outcome(ref,lambda:p(ref)); y=outcome(own,lambda:p(own))
        if x!=y:
            bad[name]=bad.get(name,0)+1
            if bad[name]<=2: print(name,sf,sb,"\n  ref",str(x)[:300],"\n  own",str(y)[:300])
print("cases",N,"differences",bad)
