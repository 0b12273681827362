#!/usr/bin/env python3
"""Long-read (-pacbio) counterpart of tools/fuzz_vs_reference.py: 60 reads of 300..9000 bases at 5..25 % error per seed, IUPAC codes,
lower case, N / n runs; host pipeline (CPU oracle backend) vs oracle/_ref/kart -t 1.  usage: python tools/fuzz_pacbio_vs_reference.py <first> <last>"""
import subprocess, os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AMD=os.environ.get('KART_FUZZ_BIN', R+'/tests/_build/kart-host-oracle')   # KART_FUZZ_BIN=<repo>/kart_amd/bin/kart-amd on a GPU box
sys.path.insert(0,R)
from kart_amd import synth
from kart_amd.index_build import read_fasta
g={n:s for n,_,s in read_fasta(R+'/tests/golden/small.fa')}
alphabet=np.frombuffer(b'ACGTNacgtnRYKMSWBDHVrykm',np.uint8)
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    rng=np.random.default_rng(seed)
    rl=int(rng.choice([300,1200,3000,7000,9000]))
    names,reads=synth.simulate_long_reads(g,60,seed=seed,read_len=rl,err=float(rng.choice([0.05,0.15,0.25])),indel_err_frac=float(rng.choice([0.1,0.5])))
    out=[]
    for r in reads:
        r=r.copy()
        k=rng.random()
        if k<0.3:
            m=rng.random(len(r))<0.02; r[m]=alphabet[rng.integers(0,len(alphabet),int(m.sum()))]
        elif k<0.4: r=np.frombuffer(r.tobytes().lower(),np.uint8).copy()
        elif k<0.5 and len(r)>200: r[100:100+int(rng.integers(1,60))]=ord('n' if rng.random()<0.5 else 'N')
        out.append(r)
    fasta = seed%3==0
    fn='pb.fa' if fasta else 'pb.fq'
    if fasta: synth.write_fasta(fn,{n:r for n,r in zip(names,out)}) if hasattr(synth,'write_fasta') else None
    else: synth.write_fastq(fn,names,out)
    res=[]
    for exe,t in ((R+'/oracle/_ref/kart','1'),(AMD,'3')):
        o='pb_%s.sam'%os.path.basename(exe)
        if os.path.exists(o): os.remove(o)
        try:
            r=subprocess.run([exe,'-silent','-pacbio','-t',t,'-i',R+'/tests/golden/idx/small','-f',fn,'-o',o],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,timeout=300)
            res.append((r.returncode, open(o,'rb').read() if os.path.exists(o) else None))
        except subprocess.TimeoutExpired: res.append(('timeout',None))
    if res[0]!=res[1]:
        bad+=1; print('DIFF seed',seed,'len',rl,'rc',res[0][0],res[1][0])
        if res[0][1] and res[1][1]:
            a=res[0][1].split(b'\n'); b=res[1][1].split(b'\n')
            for i,(x,y) in enumerate(zip(a,b)):
                if x!=y: print('  line',i,'\n   ref',x[:150],'\n   amd',y[:150]); break
print('done, diffs:',bad)
