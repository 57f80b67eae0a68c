import os, sys, gzip, tempfile, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from mapcaller_amd import api
from conftest import sam_diff
def unpack(name):
    d = tempfile.mkdtemp()
    src = os.path.join(ROOT, "tests", "golden", name)
    out = {}
    for fn in os.listdir(src):
        if fn.endswith(".gz"):
            open(os.path.join(d, fn[:-3]), "wb").write(gzip.open(os.path.join(src, fn)).read()); out[fn[:-3]] = os.path.join(d, fn[:-3])
    return out
combos = [{}, {"MCX_ORDER_MIN": "1", "MCX_NO_SIMPLE": "1"}, {"MCX_ORDER_MIN": "1"}, {"MCX_DP_LANE_ALWAYS": "1"}, {"MCX_ORDER_MIN": "1", "MCX_DP_LANE_ALWAYS": "1"},
          {"MCX_ORDER_MIN": "1", "MCX_RESCUE_IN_LINE": "1"}, {"MCX_ORDER_MIN": "1", "MCX_NO_TIER_OVERLAP": "1"}]
for name, alg in (("se", "ksw2"), ("var", "nw"), ("long", "ksw2"), ("toy", "ksw2")):
    f = unpack(name)
    r1 = [v for k, v in f.items() if k.startswith("r1.")][0]
    r2 = next((v for k, v in f.items() if k.startswith("r2.")), None)
    for env in combos:
        for k in ("MCX_ORDER_MIN", "MCX_NO_SIMPLE", "MCX_DP_LANE_ALWAYS", "MCX_RESCUE_IN_LINE", "MCX_NO_TIER_OVERLAP"):
            os.environ.pop(k, None)
        os.environ.update(env)
        ix = api.Index(os.path.join(ROOT, "tests", "golden", name, "idx"), device=0)
        mp = api.Mapper(ix, alg=alg, max_batch_reads=1 << 14)
        out = f["r1." + r1.rsplit(".", 1)[1]] + ".sam"
        try:
            st = mp.map_files(r1, r2, out)
            nd, ex = sam_diff(f[f"ref.{alg}.sam"], out)
            print(name, alg, env, "differing lines", nd, "simple", st["simple_pairs"], "tier1", st["tier1_pairs"], flush=True)
            if nd:
                for a, b in ex[:2]:
                    print("   ref:", a[:230]); print("   gpu:", b[:230])
        except Exception as e:
            print(name, alg, env, "ERROR", str(e)[:300], flush=True)
        mp.close(); ix.close()
