import sys, time, torch, faulthandler
faulthandler.enable()
sys.path.insert(0, ".")
import bench
dev = torch.device("cuda", 0)
L = int(sys.argv[1])
t = time.time()
print("gen...", flush=True)
codes = bench.make_large_codes(L, 3, dev)
torch.cuda.synchronize(); print("codes", codes.shape, time.time() - t, flush=True)
from kart_amd import index_build
anns = [("decoy", "(null)", 0, bench.DECOY_LEN, 0), ("chrE", "(null)", bench.DECOY_LEN, L, 0)]
t = time.time()
index_build.build_index_from_codes(codes.cpu().numpy(), anns, [], "/tmp/dbg_large", device="cuda", bucketed=(sys.argv[2] == "b"), verbose=True)
print("built", time.time() - t, flush=True)
