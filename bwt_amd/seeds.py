"""Host mirror of seeds.erl (/root/reference/src/seeds.erl:11-85): `generate_fs/3` writes the module
`fs` -- one `matches/Len` clause per way of spending exactly `Defects` edits on the first Len+Defects query bases, plus
a sliding `find_seeds/1`.  The reference never wires it into the worker path; here the generator's text comes from
libbwtgpu (bwt_seeds_generate_fs) and the generated module's behaviour runs on the GPU (bwt_fs_find_seeds): there is
no CPU implementation in this package (the oracle's restatement lives in oracle/)."""
from . import lib


def generate_fs(qsec, length, defects, path="fs.erl"):
    """seeds:generate_fs(Qsec, Len, Defects): writes `path` (the reference always writes "fs.erl" in the cwd,
    src/seeds.erl:13) and returns ok."""
    text = lib.seeds_generate_fs(qsec, length, defects)
    with open(path, "wb") as f:
        f.write(text)
    return "ok"


def fs_text(qsec, length, defects):
    """The bytes generate_fs would write."""
    return lib.seeds_generate_fs(qsec, length, defects)


class Fs:
    """The generated module: fs:find_seeds(Ref_Seq) for the (Qsec, Len, Defects) it was generated from."""

    def __init__(self, qsec, length, defects, device=0):
        self.qsec, self.length, self.defects, self.device = bytes(qsec), int(length), int(defects), device
        lib.seeds_generate_fs(self.qsec, self.length, self.defects)      # argument errors surface here, as at generation time

    def find_seeds(self, ref_seq):
        """1-based positions of the windows the scan accepts, newest first (a list, like the Erlang result)."""
        if isinstance(ref_seq, lib.Index):
            return lib.fs_find_seeds(self.qsec, self.length, self.defects, index=ref_seq).tolist()
        return lib.fs_find_seeds(self.qsec, self.length, self.defects, ref=ref_seq, device=self.device).tolist()
