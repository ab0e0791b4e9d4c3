"""bwt_amd -- MI355X (gfx950) implementation of cloudozer/BWT's seed-search / clustering /
Smith-Waterman hot path behind the reference's own interfaces.

Host-side mirror of the reference modules (same names, argument meaning, error behaviour):
    bwt_amd.fm_index   fm_index:fm_name/2, get_index/2        (src/fm_index.erl:145-156)
    bwt_amd.bwa        bwa:find_seeds/6                       (src/bwa.erl:12)
    bwt_amd.sga        sga:sga/6                              (src/sga.erl:17)
    bwt_amd.sw         sw:sw/2                                (src/sw.erl:47)
    bwt_amd.worker_bwt the `running` clause of worker_loop/10 (src/worker_bwt.erl:73-112)
All compute goes through libbwtgpu.so (include/bwt_gpu.h); there is no CPU fallback: using
bwt_amd.lib without the built library raises.
"""
__all__ = ["lib", "fm_index", "bwa", "sga", "sw", "worker_bwt"]
