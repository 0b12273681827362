"""probe: does starting child processes (subprocess.Popen) from a process that uses HIP through BOTH torch and libkart_amd.so
disturb the library's view of the device?  (tests/test_hg38_gpu.py saw kg_index_load report "no HIP device" right after four Popen calls)"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
g.build()
from kart_amd import api
order = sys.argv[1] if len(sys.argv) > 1 else "lib_first"
if order == "lib_first":
    api.load_library()
    print("lib loaded (no HIP call yet)")
import torch
x = torch.zeros(1 << 20, device="cuda")
torch.cuda.synchronize()
print("torch initialised; kg_device_count =", api.device_count())
p = subprocess.Popen(["sleep", "1"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, FOO="1"))
print("after one Popen: kg_device_count =", api.device_count())
ps = [subprocess.Popen(["sleep", "2"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, MALLOC_PERTURB_="85", GLIBC_TUNABLES="glibc.malloc.tcache_count=0")) for _ in range(4)]
print("after four Popen: kg_device_count =", api.device_count())
os.environ["KART_AMD_SA"] = "compact"
print("with KART_AMD_SA set: kg_device_count =", api.device_count())
for q in ps + [p]:
    q.wait()
print("children done: kg_device_count =", api.device_count(), "torch sum", float((x + 1).sum()))
