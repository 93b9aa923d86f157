"""Randomised soak of wg_zmpdisc_batch against the oracle (the wg_trig.h build): 3000 step sequences over five
sampling periods / preview windows / :omega values, with off-grid support times (phases the reference cannot hold are
refused on both sides).  Prints the number of mismatches (must be 0)."""
import os, sys, importlib, numpy as np, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import oraclelib as ol
from test_zmpdisc_oracle import kajita_model
import test_zmpdisc_gpu as t
wg = importlib.import_module("jrl-walkgen_amd"); wg.init(0)
pt = t.ptrig()
tot=0; bad=0; neg=0
for seed,(T,pv) in enumerate([(0.005,1.6),(0.004,1.0),(0.0025,0.8),(0.01,1.6),(0.005,0.3)]):
    m=kajita_model(); m.T=T; m.preview_time=pv; m.omega=[0.0,2.5,-1.5,0.0,4.0][seed]
    m.zmp_shift[0],m.zmp_shift[1],m.zmp_shift[2],m.zmp_shift[3]=0.015,0.012,0.017,0.011
    rng=np.random.default_rng(1000+seed)
    B,smax=600,20
    steps,n_steps,init=t.random_fleet(rng,B,smax,m)
    # off-grid support times: phases that do not fit their sample count exactly
    for b in range(B):
        for i in range(int(n_steps[b])):
            s=steps[b*smax+i]
            if rng.random()<0.3:
                s.ss_time=float(rng.uniform(0.3,0.9)); s.ds_time=float(rng.uniform(0.003,0.3))
    lens=[wg.zmpdisc_length(m, t.gait_steps(steps,b,smax,int(n_steps[b]))) for b in range(B)]
    lcap=max(max(lens),1)
    r=wg.zmpdisc_batch(m,steps,n_steps,init,smax,lcap)
    for b in range(B):
        o=ol.zmpdisc(m,t.gait_steps(steps,b,smax,int(n_steps[b])),init[b],lib=pt)
        tot+=1
        if o['length']<0 or r['length'][b]<0:
            neg+=1
            if (o['length']<0)!=(r['length'][b]<0): bad+=1; print('code mismatch',seed,b,o['length'],r['length'][b])
            continue
        L=o['length']
        ok = r['length'][b]==L and all(np.array_equal(r[k][b,:L],o[k]) for k in t.KEYS_D+t.KEYS_I)
        if not ok: bad+=1; print('MISMATCH',seed,b)
    print('config',seed,'done; total',tot,'refused',neg,'bad',bad, flush=True)
print('TOTAL',tot,'bad',bad)
