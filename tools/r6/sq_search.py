#!/usr/bin/env python3
"""Round-6 tool (CPU, oracle only): search for decoder states in which the certificate's conditions BIND.

For random BG1 / Zc 16 / 15-row code blocks near the waterfall that decode correctly after k iterations, one information LLR is moved against
its bit (bisection on the oracle's decoder) until that element's posterior after iteration k is 0 < rho < 1e-8.  Where every parity check
still passes at k and the bit has flipped after iteration k + 1 (the fixed schedule's last one with numIter = k + 1), the certificate is
evaluated with all conditions and without (S): a state where the latter certifies is saved (tests/golden/cert_witness_S.npz is the one of
seed 14, trial 201).

    python tools/r6/sq_search.py SEED TRIALS
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from oracle import coding as oc, certificate as cert
bgn,zc,rows=1,16,15
ils=[i for i,zs in enumerate(oc.LIFTING_SETS) if zc in zs][0]
kb=22; K=kb*zc; n_tx=(kb+4-2+rows-4)*zc
seed=int(sys.argv[1]); N=int(sys.argv[2])
rng=np.random.default_rng(seed)
found=0; tried=0; t0=time.time()
for trial in range(N):
    sigma=rng.choice([0.68,0.72,0.76,0.8])
    info=rng.integers(0,2,(1,K)).astype(np.int8)
    coded=oc.encode(info,bgn,ils,zc)
    bits=coded[:,:n_tx].astype(np.float64)
    base=np.zeros(coded.shape); base[:,:n_tx]=(2/sigma**2)*((1-2*bits)+sigma*rng.standard_normal(bits.shape))
    k=int(rng.choice([6,8,10,12]))
    b0=oc.decode(base,bgn,ils,zc,num_iter=k,rows=rows,only_info=False,belief=True)
    hard=(b0[0]<0).astype(np.int8)
    if not (hard[:K]==info[0]).all(): continue     # not converged (to the right word) at k
    for rep in range(3):
        pos=int(rng.integers(0,n_tx-4*zc if False else K-2*zc))      # an information element that was transmitted
        sgn=1-2*bits[0,pos]
        def post(L,it):
            x=base.copy(); x[0,pos]=L*sgn
            bel=oc.decode(x,bgn,ils,zc,num_iter=it,rows=rows,only_info=False,belief=True)
            return bel[0,pos+2*zc]*sgn, bel, x
        lo,hi=-60*abs(base).max(),0.0
        if post(lo,k)[0]>0 or post(hi,k)[0]<0: continue
        best=None
        for it in range(70):
            mid=0.5*(lo+hi); f,bel,x=post(mid,k)
            if f>0: hi=mid; best=(mid,f,bel,x)
            else: lo=mid
            if 0<f<1e-8: break
        if best is None: continue
        mid,f,bel,x=best
        hk=(bel[0]<0)
        if not (hk[:K]==info[0].astype(bool)).all(): continue      # the perturbed block must still pass at k
        tried+=1
        NI=k+1
        fin=oc.decode(x,bgn,ils,zc,num_iter=NI,rows=rows,only_info=False,belief=True)
        changed=((fin[0,:K]<0)!=hk[:K]).any()
        traj=[post(mid,kk)[0] for kk in (k+1,k+2,k+4,k+8,50)]
        print('trial',trial,'sigma',sigma,'k',k,'pos',pos,'rho_k %.3g'%f,'later',['%.3g'%v for v in traj],'CHANGED' if changed else '')
        if changed:
            r1=cert.decode_certified(x,bgn,ils,zc,NI,rows,[k],(),flags=1,sweeps=12)
            r0=cert.decode_certified(x,bgn,ils,zc,NI,rows,[k],(),flags=0,sweeps=12)
            print('   bits differ at',np.nonzero((r1['bits_at'][k]!=r1['bits'])[0])[0],'pos',pos+2*zc)
            print('   WITNESS? cert without S',r1['cert'][k],'full',r0['cert'][k],{n:bool(v[0]) for n,v in r0['why'][k].items()})
            found+=1
            if r1['cert'][k][0] and not r0['cert'][k][0]:
                np.savez_compressed('witness_S_seed%d_trial%d.npz'%(seed,trial), llr=x, k=k, num_iter=NI, pos=pos, info=info, bgn=bgn, zc=zc, rows=rows, rho_k=f)
                print('   SAVED')
print('tried',tried,'changed',found,'%.0f s'%(time.time()-t0))
