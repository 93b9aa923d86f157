import ctypes as C, importlib, os, sys, numpy as np
ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,os.path.join(ROOT,'tests')); sys.path.insert(0,ROOT)
import herdt_replay as hr, oraclelib as ol
import test_tick_gpu as T
wg=T._wg(); pt=T._ptrig()
model=wg.model_defaults(); wg.mpc_configure(model)
B=96
states,rng=T._random_gaits(wg,model,B,20100)
ref=(wg.GaitState*B)(); C.memmove(ref,states,C.sizeof(states))
def flat(st,prefix=""):
    out=[]
    for name,typ in st._fields_:
        v=getattr(st,name)
        if isinstance(v,C.Structure): out+=flat(v,prefix+name+".")
        elif hasattr(v,'__len__'):
            for i,e in enumerate(v):
                if isinstance(e,C.Structure): out+=flat(e,prefix+"%s[%d]."%(name,i))
                else: out.append((prefix+"%s[%d]"%(name,i),e))
        else: out.append((prefix+name,v))
    return out
for tick in range(60):
    if tick%25==0:
        for g in range(B):
            v=[rng.uniform(-0.1,0.3),rng.uniform(-0.1,0.1),rng.uniform(-0.2,0.2)]
            for st in (states[g],ref[g]): st.vref[0],st.vref[1],st.vref[2]=v
    adv=1 if tick==0 else (19 if tick==1 else 20)
    before=(wg.GaitState*B)(); C.memmove(before,ref,C.sizeof(ref))
    outs,diag,hist,hlen=wg.mpc_tick_batch(states,want_out=True,advance_calls=adv,hist_cap=256)
    bad=False
    for g in range(B):
        c=ref[g].clock
        for _ in range(adv): c+=model.Tctrl
        ref[g].clock=c
        out_c=wg.TickOut(); dump=hr.QpDump()
        pt.wgo_mpc_tick(C.byref(model),C.byref(ref[g]),C.byref(out_c),C.byref(dump))
        if T._bytes(states[g])!=T._bytes(ref[g]) or T._bytes(outs[g])!=T._bytes(out_c):
            fa,fb=flat(states[g]),flat(ref[g])
            diffs=[(n,a,b) for (n,a),(_,b) in zip(fa,fb) if a!=b and not (a!=a and b!=b)]
            oa,ob=flat(outs[g]),flat(out_c)
            odiffs=[(n,a,b) for (n,a),(_,b) in zip(oa,ob) if a!=b and not (a!=a and b!=b)]
            print("tick",tick,"gait",g,"state diffs",diffs[:8],"\n  out diffs",odiffs[:8])
            print("  diag gpu",list(diag[g]),"cpu",[dump.ifail,dump.n_iter,dump.nact,dump.n,dump.m], "hist gpu",list(hist[g,:hlen[g]]),"cpu",list(dump.hist[:dump.hist_len]))
            print("  vref",list(before[g].vref),"phase",before[g].phase,"foot",before[g].foot,"clock",ref[g].clock,"TL",before[g].time_limit)
            bad=True; break
    if bad: break
print("done")
