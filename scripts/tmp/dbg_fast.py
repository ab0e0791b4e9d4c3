import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import synth_reads
from bwt_amd import lib
from oracle import orc, erl_literal as E
length = 103
rng = np.random.default_rng(91)
g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3_000_000, dtype=np.uint8)].tobytes())
seg = bytes(g[1000:1300])
for k in range(1, 9):
    g[200_000 * k:200_000 * k + 300] = seg
ref = bytes(g)
ix = lib.Index.build(ref, shift=3)
ix.set_ktable(15, 1)
reads = (synth_reads(ref, 4000, length, 601, max_subs=2) + synth_reads(ref, 300, length, 602, indel_frac=1.0) +
         synth_reads(ref, 100, length, 603, n_frac=1.0) + synth_reads(ref, 60, length, 604, iupac_frac=1.0) +
         [ref[1000 + 7 * i:1000 + length + 7 * i] for i in range(25)] + [ref[:length], ref[5:5 + length], ref[-length:]])
order = rng.permutation(len(reads))
reads = [reads[i] for i in order]
L, SA = ix.export(); ptr, shift = ix.meta()
ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
print("oracle sga of read 21:", orc.sga(ofm, reads[21]))
for i in range(20, 25): print(i, orc.sga(ofm, reads[i]))
os.environ["BWT_SEED_DBG_GROUP"] = "4"
os.environ["BWT_DEBUG_HANDBACK"] = "1"
b = lib.Batch(ix); b.set_reads(*lib.pack_reads(reads)); c = b.run(lib.STAGE_EXTEND)
print(c["n_handed_back"])
