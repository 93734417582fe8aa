"""one seed of tools/soak_chains.py: python tools/chain_one.py <seed> [<seed> ...]   (OFL_LIB picks the library)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import soak_chains as S
import oflibnumpy_amd as of
from oracle import np_oracle as O
import test_gpu_chains as C
of.native.ensure_device(); O.build()
for seed in map(int, sys.argv[1:]):
    n, b, m = S.one_case(of, O, C, seed, 120, 160)
    print(json.dumps({"lib": os.path.basename(os.environ.get("OFL_LIB", "default")), "seed": seed, "nodes": n, "bad": b, "msgs": m}))
