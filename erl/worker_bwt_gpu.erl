%% Drop-in for the two hot clauses of worker_bwt:worker_loop/10 (reference: src/worker_bwt.erl:41-57
%% and :73-112).  Same messages in, same casts out; sga:sga/6 + sw:sw/2 per read are replaced by one
%% bwt_nif:align_batch/2 per QseqList.  Written from scratch; see INTEGRATION.md for the 12-line diff
%% that wires it into the reference's worker_bwt.erl.
-module(worker_bwt_gpu).
-export([init_index/2, running/4]).

%% replaces worker_bwt.erl:48-55: the index lives in HBM, Ref1 stays on the device as well
init_index(Chromosome, Device) ->
    {ok, FmIndices} = application:get_env(worker_bwt_app, fm_indices),
    Base = filename:join(FmIndices, Chromosome ++ "_p1"),          %% !Hardcoded part 1, as the reference
    {ok, Index} = bwt_nif:index_open(Base ++ ".fm", Base ++ ".ref", Device),
    Index.

%% replaces worker_bwt.erl:73-106 for one QseqList
running(Index, QseqList, MNode, MPid) ->
    Results = bwt_nif:align_batch(Index, QseqList),
    %% the reference answers reads in reverse batch order (Seeds is built by prepending, :74-80)
    lists:foreach(fun
        ({_Name, nothing}) -> ok;                                   %% sga returned [] (:77)
        ({Name, {error, Reason}}) ->
            lager:error("read ~s: the reference would have crashed here: ~p", [Name, Reason]);
        ({_Name, []}) ->
            navel:call_no_return(MNode, gen_server, cast, [MPid, no_cigar]);
        ({Name, [{Cigar, P, RefSeq}]}) ->
            navel:call_no_return(MNode, gen_server, cast, [MPid, {cigar, Name, Cigar, P, RefSeq}]);
        ({Name, Cigars}) ->
            %% the very same lists:sort/2 call as worker_bwt.erl:104: OTP decides ties
            [{TopCigar, P, RefSeq} | _] =
                lists:sort(fun({{R1,_},_,_}, {{R2,_},_,_}) -> R1 > R2 end, Cigars),
            navel:call_no_return(MNode, gen_server, cast, [MPid, {cigar, Name, TopCigar, P, RefSeq}])
    end, lists:reverse(Results)).
