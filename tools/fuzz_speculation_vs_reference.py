#!/usr/bin/env python3
"""Paired-end files whose insert-size distribution changes along the file (2-5 segments of 4-30 k pairs, means 180..1600), so that the
running EstDistance keeps moving and many speculated chunks have to be re-mapped at commit; host pipeline at 2..16 threads vs
oracle/_ref/kart -t 1.  usage: python tools/fuzz_speculation_vs_reference.py <first_seed> <last_seed>   (KART_FUZZ_BIN as in fuzz_vs_reference.py)"""
import subprocess, os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0,R)
from kart_amd import synth
from kart_amd.index_build import read_fasta
g={n:s for n,_,s in read_fasta(R+'/tests/golden/small.fa')}
AMD=os.environ.get('KART_FUZZ_BIN', R+'/tests/_build/kart-host-oracle')
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    parts=[]
    n_total=0
    # the insert-size distribution changes along the file, so the running EstDistance keeps moving
    for seg in range(int(rng.integers(2,6))):
        n=int(rng.integers(4000,30000)); ins=float(rng.choice([180,250,320,450,700,1200,1600])); sd=ins/float(rng.choice([4,8,20]))
        names,r1,r2=synth.simulate_pairs(g,n,seed=seed*10+seg,err=float(rng.choice([0.005,0.02,0.05])),mut=0.002,indel_frac=0.3,ins_mean=ins,ins_sd=sd)
        parts.append((names,r1,r2)); n_total+=n
    names=['%s_%d'%(nm,i) for i,p in enumerate(parts) for nm in p[0]]
    r1=[x for p in parts for x in p[1]]; r2=[x for p in parts for x in p[2]]
    synth.write_fastq('sp1.fq',names,r1,mate=1); synth.write_fastq('sp2.fq',names,r2,mate=2)
    res=[]
    threads=int(rng.choice([2,5,8,16]))
    for exe,t in ((R+'/oracle/_ref/kart','1'),(AMD,str(threads))):
        o='sp_%s.sam'%os.path.basename(exe)
        if os.path.exists(o): os.remove(o)
        r=subprocess.run([exe,'-silent','-t',t,'-i',R+'/tests/golden/idx/small','-f','sp1.fq','-f2','sp2.fq','-o',o],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,env=dict(os.environ,KART_AMD_VERBOSE='1'))
        res.append((r.returncode, open(o,'rb').read()))
        if exe==AMD: resp=[l for l in r.stdout.decode().splitlines() if 're-mapped' in l]
    same=res[0]==res[1]
    print('seed',seed,'pairs',n_total,'threads',threads,'same' if same else 'DIFF',resp)
    bad+=0 if same else 1
print('done, diffs:',bad)
