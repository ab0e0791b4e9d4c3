%% Loader of the NIF shim c_src/bwt_nif.c (libbwtgpu, include/bwt_gpu.h).
%% Written from scratch for this repository; not compiled here (no Erlang in the authoring container).
-module(bwt_nif).
-export([constants/0, index_open/3, index_meta/1, align_batch/2, align_workload/2]).
-on_load(init/0).

-include("bwt.hrl").   %% the reference's own include/bwt.hrl

init() ->
    So = filename:join(code:priv_dir(bwt), "bwt_nif"),
    ok = erlang:load_nif(So, 0),
    %% the kernels are compiled with these constants: refuse to run against a different bwt.hrl
    {?TOLERANCE, ?MIN_LEN, ?MAX_RANGE, ?MAX_LEN, ?MATCH, ?MISMATCH, ?GAP_PENALTY, ?GAP_EXT_PENALTY, ?UNKNOWN, 200} =
        constants(),
    ok.

constants() -> erlang:nif_error(nif_not_loaded).
index_open(_Fm, _Ref, _Device) -> erlang:nif_error(nif_not_loaded).
index_meta(_Index) -> erlang:nif_error(nif_not_loaded).
align_batch(_Index, _QseqList) -> erlang:nif_error(nif_not_loaded).
align_workload(_Index, _Workload) -> erlang:nif_error(nif_not_loaded).
