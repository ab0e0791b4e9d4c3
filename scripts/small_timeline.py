"""One list of the reference's 200 reads (and one whole message of 2000) through bwt_small_align, a few dozen times: run under
`rocprofv3 --kernel-trace` this gives the device's own timeline of a call (begin / end of every kernel, the gaps between them).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/small_tl -- python3 scripts/small_timeline.py [n] [small_graph]"""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
from bwt_amd import lib     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
graph = int(sys.argv[2]) if len(sys.argv) > 2 else -1
g = os.path.join(ROOT, "tests", "golden", "fm_indices", "GL000193.1_p1")
ix = lib.Index.open(g + ".fm", g + ".ref")
ref = np.frombuffer(open(g + ".ref", "rb").read(), np.uint8)
rng = np.random.default_rng(1)
st = rng.integers(0, len(ref) - 100, n)
bases = ref[st[:, None] + np.arange(100)[None, :]].reshape(-1).copy()
k = rng.integers(0, 3, n)
for s in range(2):
    rows = np.nonzero(k > s)[0]
    bases.reshape(n, 100)[rows, rng.integers(0, 100, len(rows))] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(rows))]
off = np.arange(0, (n + 1) * 100, 100, dtype=np.uint64)
sb = lib.SmallBatch(ix)
sb.set_options(small_graph=graph)
for _ in range(40):
    sb.align_raw(bases, off)
print(sb.stage_us(), sb.graphs)
sb.close()
