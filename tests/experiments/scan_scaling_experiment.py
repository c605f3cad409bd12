#!/usr/bin/env python3
"""Experiment (CPU, numpy): does balancing the state coordinates (x = D x~, D from the averaged diagonals of C and J) repair the accuracy
of the parallel-scan backward sweep on the WHOLE-BODY stage QPs?  Result recorded in DESIGN.md: cond(I + C1 J2) drops from ~1e9 to
1e6-1e7, the disagreement with the serial recursion does not (5e-8 -> 3e-8): the loss is not the conditioning of the solve alone."""
import sys, ctypes as C, os, math
import os; _R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [_R, os.path.join(_R, 'oracle'), os.path.join(_R, 'tests')]
import numpy as np
import parallel_scan as ps
import test_parallel_scan as T
from test_oracle_lq import perturbed_problem
from wb_humanoid_mpc_amd import load_model, _abi
m=load_model()
lib=C.CDLL(os.path.join(_R, 'tests', 'hostemu', 'libhsqp_hostemu.so')); lib.emu_create.restype=C.c_void_p
err=C.create_string_buffer(256); h=C.c_void_p(lib.emu_create(C.byref(m.desc),err,256))
P=T.P
def run(gait,n,seed=5):
    x0,x,u,par,dt=perturbed_problem(m,n,gait,seed=seed)
    xn,un,dx,du=np.zeros_like(x),np.zeros_like(u),np.zeros_like(x),np.zeros_like(u)
    kkt,pb,pa=np.zeros(2),np.zeros(3),np.zeros(3); qp=np.zeros((n,lib.emu_qp_size()))
    assert lib.emu_sqp_iteration(h,n,C.c_double(dt),P(x0),P(x),P(u),P(par),P(xn),P(un),P(dx),P(du),P(kkt),P(pb),P(pa),P(qp))==0
    st=T.stages_from_records(qp,58)
    Qf=np.array(m.raw['Qf']); qN=Qf*(x[n]-par[n,:58]); dx0=(x0-x[0])
    return st,np.diag(Qf),qN,dx0,dx,du,qp
def scaled_solve(st,QN,qN,dx0,D):
    Di=1.0/D
    st2=[dict(A=(Di[:,None]*s['A'])*D[None,:],B=Di[:,None]*s['B'],b=Di*s['b'],Q=(D[:,None]*s['Q'])*D[None,:],P=s['P']*D[None,:],R=s['R'],q=D*s['q'],r=s['r']) for s in st]
    sdx,sut,S,sv,lv=ps.solve_qp(st2,(D[:,None]*QN)*D[None,:],D*qN,Di*dx0)
    return sdx*D[None,:],sut
conds=[]
orig=ps.combine
def comb(e1,e2):
    conds.append(np.linalg.cond(np.eye(e1[0].shape[0])+e1[2]@e2[4])); return orig(e1,e2)
ps.combine=comb
for gait,n in (('walk',20),('run',33),('walk',100)):
    st,QN,qN,dx0,dx,du,qp=run(gait,n)
    sc=max(1,np.abs(dx).max(),np.abs(du).max())
    conds.clear(); sdx,sut=scaled_solve(st,QN,qN,dx0,np.ones(58)); e0=np.abs(sdx-dx).max()/sc; c0=max(conds)
    # balance from averaged diagonals of single-stage elements
    els=[ps.stage_element(**s) for s in st]
    Cd=np.mean([np.abs(np.diag(e[2])) for e in els],axis=0)+1e-300; Jd=np.mean([np.abs(np.diag(e[4])) for e in els],axis=0)+np.diag(QN)+1e-300
    D=(Cd/Jd)**0.25
    D=np.clip(D, 1e-6, 1e6)
    conds.clear(); sdx2,sut2=scaled_solve(st,QN,qN,dx0,D); e1=np.abs(sdx2-dx).max()/sc; c1=max(conds)
    print(gait,n,'unscaled err %.1e cond %.1e | balanced err %.1e cond %.1e'%(e0,c0,e1,c1), 'D range %.1e..%.1e'%(D.min(),D.max()))
