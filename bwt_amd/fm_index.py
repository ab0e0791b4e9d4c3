"""Host mirror of fm_index:fm_name/2 and fm_index:get_index/2
(/root/reference/src/fm_index.erl:145-156) plus the worker's index acquisition
(/root/reference/src/worker_bwt.erl:48-55): the index lives in HBM behind a handle instead of
an Erlang tuple, Meta keeps the reference's proplist shape."""
import os

from . import lib

FM_FOLDER = "fm_indices/"        # fm_index.erl:12
_env = {}


def set_env(key, value):
    """application:set_env(_, fm_indices, Path) analogue (fm_index.erl:146, worker_bwt.erl:51)."""
    _env[key] = value


def fm_name(chrom, j):
    path = _env.get("fm_indices")
    if path is not None:
        return os.path.join(path, chrom) + "_p" + str(j)      # filename:join(Path,Chrom)++"_p"++J
    return FM_FOLDER + chrom + "_p" + str(j)


class FM:
    """What worker_loop keeps as {FM, Ref1, Pc,Pg,Pt,Last, Shift}: a device-resident index."""

    def __init__(self, index):
        self.index = index
        (self.Pc, self.Pg, self.Pt, self.Last), self.shift = index.meta()

    def close(self):
        self.index.close()


def get_index(chrom, j, device=0):
    """-> (Meta, FM) with Meta = [("pointers",(Pc,Pg,Pt,Last)), ("shift",Shift)].
    Raises FileNotFoundError / lib.BwtError where the reference's {ok,Bin} = file:read_file badmatches."""
    base = fm_name(chrom, j)
    if not os.path.exists(base + ".fm"):
        raise FileNotFoundError(base + ".fm")
    fm = FM(lib.Index.open(base + ".fm", base + ".ref", device))
    meta = [("pointers", (fm.Pc, fm.Pg, fm.Pt, fm.Last)), ("shift", fm.shift)]
    return meta, fm
