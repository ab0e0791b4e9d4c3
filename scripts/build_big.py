import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bwt_amd import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
rng = np.random.default_rng(3)
t=time.time(); ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]; print("gen %.1fs"%(time.time()-t), flush=True)
t=time.time(); ix = lib.Index.build(ref, shift=1); dt=time.time()-t
print("build n=%d: %.2fs  meta=%s bytes=%s" % (n, dt, ix.meta(), ix.device_bytes()), flush=True)
