#!/usr/bin/env python3
"""Differential fuzz against the imported reference (build container only, no GPU): see the probe table below.
Reads stay inside the streams (the reference's own generators raise RuntimeError at the end of a finite stream on Python >= 3.7)."""
import sys, random, warnings, operator, itertools
warnings.filterwarnings("ignore")
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/reference')
import audiolazy as ref, audiolazy_amd as own

def norm(v):
    if isinstance(v,bool) or v is None: return v
    if isinstance(v,int): return ("i",v)
    if isinstance(v,float): return ("f",v.hex())
    if isinstance(v,complex): return ("c",v.real.hex(),v.imag.hex())
    if isinstance(v,(list,tuple)): return [norm(x) for x in v]
    if isinstance(v,str): return ("s",v)
    if hasattr(v,'take'): return ("S",[norm(x) for x in v.take(3)])
    if hasattr(v,'__iter__'): return ("it",[norm(x) for x in itertools.islice(v,3)])
    return ("?",type(v).__name__)
def outcome(fn):
    try: return norm(fn())
    except Exception as e: return ("raises",type(e).__name__)
rng=random.Random(3)
def data():
    k=rng.choice(["ints","floats","mixed"])
    n={"short":2,"empty":0}.get(k,rng.randint(4,9))
    if k=="ints": return [rng.randint(-4,4) for _ in range(n)]
    return [rng.choice([rng.uniform(-2,2), rng.randint(-3,3), .5]) for _ in range(n)]
P={
 "add_s": lambda m,a,b: m.Stream(a)+m.Stream(b), "add_l": lambda m,a,b: m.Stream(a)+b, "radd_l": lambda m,a,b: b+m.Stream(a) if False else 2+m.Stream(a),
 "sub": lambda m,a,b: m.Stream(a)-m.Stream(b), "rsub": lambda m,a,b: 1.5-m.Stream(a), "mul": lambda m,a,b: m.Stream(a)*m.Stream(b), "rmul": lambda m,a,b: 3*m.Stream(a),
 "div": lambda m,a,b: m.Stream(a)/m.Stream(b), "rdiv": lambda m,a,b: 2/m.Stream(a), "floordiv": lambda m,a,b: m.Stream(a)//2, "mod": lambda m,a,b: m.Stream(a)%3,
 "pow": lambda m,a,b: m.Stream(a)**2, "rpow": lambda m,a,b: 2**m.Stream(a), "powh": lambda m,a,b: m.Stream(a)**.5,
 "neg": lambda m,a,b: -m.Stream(a), "pos": lambda m,a,b: +m.Stream(a), "abs": lambda m,a,b: abs(m.Stream(a)), "inv": lambda m,a,b: ~m.Stream([int(x) for x in a]),
 "lt": lambda m,a,b: m.Stream(a)<m.Stream(b), "ge": lambda m,a,b: m.Stream(a)>=0, "eq": lambda m,a,b: m.Stream(a)==m.Stream(b), "ne": lambda m,a,b: m.Stream(a)!=1,
 "and": lambda m,a,b: m.Stream([int(x) for x in a])&3, "or": lambda m,a,b: m.Stream([int(x) for x in a])|1, "xor": lambda m,a,b: m.Stream([int(x) for x in a])^5, "shl": lambda m,a,b: m.Stream([int(abs(x)) for x in a])<<1,
 "take": lambda m,a,b: m.Stream(a).take(3),
 "take1": lambda m,a,b: m.Stream(a).take(1), "take0": lambda m,a,b: m.Stream(a).take(0),
 "peek": lambda m,a,b: (lambda s:(s.peek(2),s.take(3)))(m.Stream(a)), "peek0": lambda m,a,b: m.Stream(a).peek(),
 "skip": lambda m,a,b: m.Stream(a).skip(2), "map": lambda m,a,b: m.Stream(a).map(lambda x:x*2),
 "copy": lambda m,a,b: (lambda s:(s.copy().take(2), s.take(3)))(m.Stream(a)),
 "ctor_num": lambda m,a,b: m.Stream(5).take(3), "ctor_multi": lambda m,a,b: m.Stream(1,2,3).take(7), "ctor_mixed": lambda m,a,b: m.Stream(a,3).take(5), "ctor_none": lambda m,a,b: m.Stream(),
 "getattr": lambda m,a,b: m.Stream([complex(x,1) for x in a]).real, "call": lambda m,a,b: m.Stream([abs, abs])(-2).take(2),
 "bool": lambda m,a,b: bool(m.Stream(a)), "getitem": lambda m,a,b: m.Stream(a)[1],
 "thub2": lambda m,a,b: (lambda t:(t+t).take(4))(m.thub(m.Stream(a),2)), "thub_num": lambda m,a,b: m.thub(3,2),
 "control": lambda m,a,b: (lambda c:(c.take(2), setattr(c,"value",7), c.take(2)))(m.ControlStream(1)),
 "zero_pad": lambda m,a,b: list(m.zero_pad(a,left=2,right=1)), "zero_pad_z": lambda m,a,b: list(m.zero_pad(a,left=1,zero=9)),
 "blocks_fn": lambda m,a,b: [list(x) for x in m.blocks(a,3,hop=2)], "blocks_pad": lambda m,a,b: [list(x) for x in m.blocks(a,4,padval=7)],
 "rint": lambda m,a,b: [m.rint(x) for x in a], "rint_step": lambda m,a,b: [m.rint(x,2) for x in a],
 "almost_eq": lambda m,a,b: m.almost_eq(a,[x+1e-9 for x in a]), "almost_eq_diff": lambda m,a,b: m.almost_eq.diff(a,b),
 "dB20": lambda m,a,b: list(m.dB20(a)), "dB10": lambda m,a,b: m.dB10(4.0),
 "sHz": lambda m,a,b: list(m.sHz(48000)), "freq2lag": lambda m,a,b: m.freq2lag(.3), "zeros": lambda m,a,b: m.zeros(3).take(5), "ones": lambda m,a,b: m.ones(2).take(5), "zeros_inf": lambda m,a,b: m.zeros().take(3),
}
bad={}; known={}
N=int(sys.argv[1]) if len(sys.argv)>1 else 200
for i in range(N):
    a,b=data(),data()
    for name,p in P.items():
        x=outcome(lambda:p(ref,list(a),list(b))); y=outcome(lambda:p(own,list(a),list(b)))
        if x==("raises","RuntimeError") and y[0]!="raises":
            known[name]=known.get(name,0)+1   # PEP 479: the reference's take / skip generators die at a finite stream's end on Python >= 3.7
        elif x!=y:
            bad[name]=bad.get(name,0)+1
            if bad[name]<=2: print(name,a,b,"\n  ref",str(x)[:200],"\n  own",str(y)[:200])
print("cases",N,"differences",bad,"of",len(P),"| reads past a finite stream's end (reference: RuntimeError, own: the short result)",known)
