#!/usr/bin/env python3
"""Fuzz the host pipeline (CPU oracle backend, tests/_build/kart-host-oracle) against the unmodified reference binary
(oracle/_ref/kart -t 1) on odd but legal reads: lengths 1..420, IUPAC codes, lower case, N / n runs, both strands, PE /
SE / FASTA / -m.  VALIDATION TOOL (needs oracle/_ref).  usage: python tools/fuzz_vs_reference.py <first_seed> <last_seed>
Run from a scratch directory; prints every differing case."""
import subprocess, os, sys, numpy as np
R=os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AMD=os.environ.get('KART_FUZZ_BIN', R+'/tests/_build/kart-host-oracle')   # KART_FUZZ_BIN=<repo>/kart_amd/bin/kart-amd on a GPU box
sys.path.insert(0,R)
from kart_amd.index_build import read_fasta
g={n:s for n,_,s in read_fasta(R+'/tests/golden/small.fa')}
chrA=g['chrA']; chrB=g['chrB']
comp=np.zeros(256,np.uint8); 
for a,b in zip(b'ACGTacgtNn',b'TGCAtgcaNn'): comp[a]=b
def rc(x): return comp[x[::-1]]
alphabet=np.frombuffer(b'ACGTNacgtnRYKMSWBDHVrykm',np.uint8)
def run(seed, mode):
    rng=np.random.default_rng(seed)
    r1=[];r2=[]
    for i in range(600):
        L1=int(rng.choice([rng.integers(1,40),rng.integers(40,160),150,rng.integers(160,420)]))
        L2=L1
        src=chrA if rng.random()<0.7 else chrB
        ins=int(rng.integers(max(L1,L2)+1, max(L1,L2)+600))
        p=int(rng.integers(0,len(src)-ins-1))
        a=src[p:p+L1].copy(); b=rc(src[p+ins-L2:p+ins].copy())
        if rng.random()<0.3: a,b=b,a
        for x in (a,b):
            k=rng.random()
            if k<0.3:
                m=rng.random(len(x))<rng.choice([0.01,0.05,0.2]); x[m]=alphabet[rng.integers(0,len(alphabet),int(m.sum()))]
            elif k<0.4: x[:]=np.frombuffer(x.tobytes().lower(),np.uint8)
            elif k<0.45 and len(x)>30: x[10:10+int(rng.integers(1,25))]=ord('n' if rng.random()<0.5 else 'N')
        r1.append(a); r2.append(b)
    def wr(path, reads, mate, fasta):
        with open(path,'wb') as fh:
            for i,r in enumerate(reads):
                nm=b'f%d extra/%d'%(i,mate)
                if fasta: fh.write(b'>'+nm+b'\n'+r.tobytes()+b'\n')
                else: fh.write(b'@'+nm+b'\n'+r.tobytes()+b'\n+\n'+bytes([33+int(q) for q in rng.integers(0,40,len(r))])+b'\n')
    fasta = mode=='fa'
    ext='fa' if fasta else 'fq'
    wr('z1.'+ext,r1,1,fasta); wr('z2.'+ext,r2,2,fasta)
    args={'pe':['-f','z1.fq','-f2','z2.fq'],'se':['-f','z1.fq'],'fa':['-f','z1.fa','-f2','z2.fa'],'pe_m':['-f','z1.fq','-f2','z2.fq','-m']}[mode]
    outs=[]
    for exe,t in ((R+'/oracle/_ref/kart','1'),(AMD,'3')):
        out='z_%s.sam'%os.path.basename(exe)
        if os.path.exists(out): os.remove(out)
        try:
            r=subprocess.run([exe,'-silent','-t',t,'-i',R+'/tests/golden/idx/small']+args+['-o',out],stdout=subprocess.PIPE,stderr=subprocess.STDOUT,timeout=120,
                             env=dict(os.environ, MALLOC_PERTURB_='85'))
            outs.append((r.returncode, open(out,'rb').read() if os.path.exists(out) else None))
        except subprocess.TimeoutExpired:
            outs.append(('timeout',None))
    if mode == 'pe_m' and outs[0][1] and outs[1][1]:
        # -m with pairs: when a mate is "unique" (score > sub_score) SetPairedAlignmentFlag (src/Mapping.cpp:97-103,127-135) sets
        # the FLAG of its best report only, yet every report with a score is printed: the others carry whatever the heap held
        # (SURVEY App. B-12).  This build prints 0 there -- never a legal FLAG of a paired read -- so such fields are masked.
        a = outs[0][1].split(b'\n'); b = outs[1][1].split(b'\n')
        if len(a) == len(b):
            for i, (x, y) in enumerate(zip(a, b)):
                fx, fy = x.split(b'\t', 2), y.split(b'\t', 2)
                if len(fx) == 3 and len(fy) == 3 and not y.startswith(b'@') and fy[1] == b'0':
                    a[i] = fx[0] + b'\tX\t' + fx[2]; b[i] = fy[0] + b'\tX\t' + fy[2]
            outs = [(outs[0][0], b'\n'.join(a)), (outs[1][0], b'\n'.join(b))]
    return outs
bad=0
for seed in range(int(sys.argv[1]),int(sys.argv[2])):
    for mode in ('pe','se','fa','pe_m'):
        o=run(seed,mode)
        same=o[0]==o[1]
        if not same:
            bad+=1
            print('DIFF seed',seed,mode,'rc',o[0][0],o[1][0])
            if o[0][1] and o[1][1]:
                a=o[0][1].split(b'\n'); b=o[1][1].split(b'\n')
                for i,(x,y) in enumerate(zip(a,b)):
                    if x!=y: print('  line',i,'\n   ref',x[:170],'\n   amd',y[:170]); break
print('done, diffs:',bad)
