#!/usr/bin/env python3
"""Experiment (CPU, numpy; run from the repo root): parallel single-stage Riccati sweeps after the scan on the WHOLE-BODY stage QPs.
One sweep takes the disagreement with the serial recursion from 5e-8 to 5e-9; further sweeps plateau at ~3e-9."""
import sys, ctypes as C
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [_R, os.path.join(_R, 'oracle'), os.path.join(_R, 'tests'), os.path.join(_R, 'tests', 'experiments')]
import numpy as np
import parallel_scan as ps
import scan_scaling_experiment as E   # reuses run() (prints its own table first)
def ric_step(st,Sn,sn):
    A,B,b,Q,Pm,R,q,r=(st[k] for k in ('A','B','b','Q','P','R','q','r'))
    Lam=R+B.T@Sn@B; G=Pm+B.T@Sn@A; g=r+B.T@(sn+Sn@b)
    K=-np.linalg.solve(Lam,G); kv=-np.linalg.solve(Lam,g)
    S=Q+A.T@Sn@A+G.T@K; s=q+A.T@(sn+Sn@b)+G.T@kv
    return 0.5*(S+S.T),s,K,kv
def rollout(stages,K,kv,dx0):
    dx=[dx0]; us=[]
    for k,st in enumerate(stages):
        u=K[k]@dx[-1]+kv[k]; us.append(u); dx.append(st['A']@dx[-1]+st['B']@u+st['b'])
    return np.array(dx),np.array(us)
for gait,n in (('walk',20),('walk',100)):
    st,QN,qN,dx0,dx,du,qp=E.run(gait,n)
    sc=max(1,np.abs(dx).max(),np.abs(du).max())
    els=[ps.stage_element(**s) for s in st]+[ps.terminal_element(QN,qN)]
    suf,_=ps.suffix_scan(els)
    S=[e[4] for e in suf]; s=[-e[3] for e in suf]
    for it in range(4):
        Ks,ks=[],[]
        Sn,sn=[None]*(n+1),[None]*(n+1); Sn[n],sn[n]=S[n],s[n]
        for k in range(n):
            Sk,sk,K,kv=ric_step(st[k],S[k+1],s[k+1]); Ks.append(K); ks.append(kv); Sn[k],sn[k]=Sk,sk
        sdx,sus=rollout(st,Ks,ks,dx0)
        print(gait,n,'refinement sweeps',it,'state err %.1e'%(np.abs(sdx-dx).max()/sc))
        S,s=Sn,sn
