import gzip, subprocess, os, sys
G="tests/golden/sam"; IDX="tests/golden/idx/small"
want=gzip.open(f"{G}/pe_g2.sam.gz").read().split(b"\n")
want_pe=gzip.open(f"{G}/pe.sam.gz").read().split(b"\n")
bad=0
fh=open("gpurun_out/flaky_diff.txt","w")
for it in range(int(sys.argv[1]) if len(sys.argv)>1 else 40):
    for case,extra,w in (("g2",["-g","2"],want),("pe",[],want_pe)):
        out=f"/tmp/o_{case}.sam"
        subprocess.run(["kart_amd/bin/kart-amd","-silent","-i",IDX,"-f",f"{G}/pe_1.fq.gz","-f2",f"{G}/pe_2.fq.gz","-o",out]+extra,check=True,stdout=subprocess.DEVNULL)
        got=open(out,"rb").read().split(b"\n")
        d=[(i,a,b) for i,(a,b) in enumerate(zip(got,w)) if a!=b]
        if d or len(got)!=len(w):
            bad+=1
            for i,a,b in d[:5]:
                fa,fb=a.split(b"\t"),b.split(b"\t")
                cols=[j for j,(x,y) in enumerate(zip(fa,fb)) if x!=y]
                fh.write(f"iter {it} case {case} line {i} cols {cols}\nGOT  {a.decode()}\nWANT {b.decode()}\n")
print("bad runs:",bad)
fh.close()
print(open("gpurun_out/flaky_diff.txt").read()[:4000])
