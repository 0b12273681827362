import sys, numpy as np
sys.path.insert(0, ".")
from kart_amd import api
ix = api.Index("tests/golden/idx/small", 0, api.KG_SA_SAMPLED)
rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
def mk(n):
    pairs = []
    for i in range(n):
        cls = i % 10
        hi = 9 if cls < 7 else 33 if cls < 9 else 90
        m = int(rng.integers(1, hi)); k = int(rng.integers(1, hi))
        a = alpha[rng.integers(0, 4, size=m)].tobytes()
        b = a if i % 3 == 0 else alpha[rng.integers(0, 4, size=k)].tobytes()
        pairs.append((a, b))
    return pairs
for n in (300, 3000, 30000):
    pairs = mk(n)
    ref = None
    bad = 0
    for it in range(40):
        ops = ix.nw_ops(pairs)
        key = [o.tobytes() for o in ops]
        if ref is None: ref = key
        else:
            d = [i for i, (x, y) in enumerate(zip(ref, key)) if x != y]
            if d:
                bad += 1
                i = d[0]
                print("n", n, "iter", it, "ndiff", len(d), "first", i, len(pairs[i][0]), len(pairs[i][1]), "len", len(ref[i]), len(key[i]), "classes", sorted(set(max(len(pairs[j][0]), len(pairs[j][1])) for j in d))[:20])
    print("n", n, "bad iterations", bad)
