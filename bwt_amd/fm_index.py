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
        sb = getattr(self, "_small_batch", None)          # (worker_bwt's small-batch handle holds device buffers on this index)
        if sb is not None:
            sb.close()
            self._small_batch = None
        self.index.close()

    def save_native(self, chrom, j):
        """Keeps the index as <fm_indices>/<Chrom>_p<J>.bwtgpu (bwt_index_save_native); get_index finds it when no .fm exists."""
        self.index.save_native(fm_name(chrom, j) + ".bwtgpu")


def get_index(chrom, j, device=0):
    """-> (Meta, FM) with Meta = [("pointers",(Pc,Pg,Pt,Last)), ("shift",Shift)].
    Raises FileNotFoundError / lib.BwtError where the reference's {ok,Bin} = file:read_file badmatches."""
    base = fm_name(chrom, j)
    if not os.path.exists(base + ".fm"):
        if os.path.exists(base + ".bwtgpu"):             # the compact native file (monolithic indexes: no .fm of that size exists)
            fm = FM(lib.Index.open_native(base + ".bwtgpu", device))
            return [("pointers", (fm.Pc, fm.Pg, fm.Pt, fm.Last)), ("shift", fm.shift)], fm
        raise FileNotFoundError(base + ".fm")
    fm = FM(lib.Index.open(base + ".fm", base + ".ref", device))
    meta = [("pointers", (fm.Pc, fm.Pg, fm.Pt, fm.Last)), ("shift", fm.shift)]
    return meta, fm


# ----------------------------------------------------------------------------------------------
# Offline index construction: fm_index:make_indices/1,2 (/root/reference/src/fm_index.erl:23-141)
OVERLAP = 2                      # fm_index.erl:11


def _ltrim_N(ls):
    """fm_index.erl:172-177: drop leading N / LF / CR, counting only the N."""
    j = k = 0
    while k < len(ls):
        c = ls[k]
        if c == 0x4E:
            j += 1
        elif c not in (10, 13):
            break
        k += 1
    return j, ls[k:]


def _contigs(genome_file):
    """Yields (name, raw bytes between this header line and the next) -- find_chromosome /
    read_chromosome (fm_index.erl:71-90): name = first token of the header, split on ' ' and LF."""
    with open(genome_file, "rb") as f:
        data = f.read()
    pos = 0
    name = None
    start = 0
    while pos < len(data):
        nl = data.find(b"\n", pos)
        end = len(data) if nl < 0 else nl + 1
        line = data[pos:end]
        if line[:1] == b">":
            if name is not None:
                yield name, data[start:pos]
            toks = [t for t in line[1:].replace(b"\n", b" ").split(b" ") if t]
            name = toks[0].decode()
            start = end
        pos = end
    if name is not None:
        yield name, data[start:]


def make_indices(genome_file, chromo_ls="all", out_dir=None, device=0, rng_state=None):
    """Builds <Chrom>_p<J>.fm / .ref for the listed contigs exactly as the reference lays them out:
    flank-N trim and Shift (fm_index.erl:102-105), CR/LF strip (:109), chunk count
    Nbr = 1 + ((Len+2) bsr 24) and 2-bp overlap (:113,121-123), Meta shift = Shift+Start (:123).
    The AS183 state for N bases is threaded through all chunks like the Erlang process state.
    Returns the list of base paths written."""
    if rng_state is None:
        rng_state = [3172, 9814, 20125]
    if out_dir is not None:
        set_env("fm_indices", out_dir)
    written = []
    for name, raw in _contigs(genome_file):
        if chromo_ls != "all" and name not in chromo_ls:
            continue
        _, rev = _ltrim_N(raw[::-1])
        shift, chromo2 = _ltrim_N(rev[::-1])
        body = bytes(c for c in chromo2 if c not in (10, 13))
        nbr = 1 + ((len(body) + OVERLAP) >> 24)
        for j in range(1, nbr + 1):
            chunk_size = (len(body) + nbr * OVERLAP) // nbr
            start = 1 + (j - 1) * (chunk_size - OVERLAP)
            chunk = body[start - 1:start - 1 + chunk_size]                 # lists:sublist(Body,Start,Chunk_size)
            base = fm_name(name, j)
            ix = lib.Index.build(chunk, shift + start, rng_state, device)
            try:
                ix.save(base + ".fm", base + ".ref")
            finally:
                ix.close()
            written.append(base)
    return written
