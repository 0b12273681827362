import gzip, subprocess, os
G="tests/golden/sam"; IDX="tests/golden/idx/small"
out="/tmp/o.sam"
subprocess.run(["kart_amd/bin/kart-amd","-silent","-i",IDX,"-f",f"{G}/pe_1.fq.gz","-f2",f"{G}/pe_2.fq.gz","-g","2","-o",out],check=True,stdout=subprocess.DEVNULL)
got=open(out,"rb").read().split(b"\n"); want=gzip.open(f"{G}/pe_g2.sam.gz").read().split(b"\n")
with open("gpurun_out/g2_diff.txt","w") as fh:
    for i,(a,b) in enumerate(zip(got,want)):
        if a!=b:
            fh.write(f"line {i}\nGOT  {a.decode()}\nWANT {b.decode()}\n")
print(open("gpurun_out/g2_diff.txt").read()[:3000])
