import os, sys, gzip, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MCX_ORDER_MIN"] = "1"; os.environ["MCX_TIMING"] = "1"
from mapcaller_amd import api
d = tempfile.mkdtemp()
src = os.path.join(ROOT, "tests", "golden", "toy")
for fn in ("r1.fq.gz", "r2.fq.gz"):
    open(os.path.join(d, fn[:-3]), "wb").write(gzip.open(os.path.join(src, fn)).read())
ix = api.Index(os.path.join(src, "idx"), device=0)
mp = api.Mapper(ix, alg="ksw2", max_batch_reads=1 << 14)
st = mp.map_files(os.path.join(d, "r1.fq"), os.path.join(d, "r2.fq"), os.path.join(d, "o.sam"))
print({k: st[k] for k in ("reads", "mapped", "dp_jobs", "tier1_pairs", "simple_pairs", "replayed_pairs")})
