%% Drop-in for the two hot clauses of worker_bwt:worker_loop/10 (reference: src/worker_bwt.erl:41-57
%% and :73-112).  Same messages in, same casts out; sga:sga/6 + sw:sw/2 per read are replaced by one
%% bwt_nif:align_batch/2 per QseqList.  Written from scratch; see INTEGRATION.md for the 12-line diff
%% that wires it into the reference's worker_bwt.erl.
%%
%% `backend` (app env worker_bwt_app, default gpu; SURVEY section 5 "Config/flags"):
%%   gpu     the index lives in HBM, one NIF call per QseqList;
%%   erlang  no GPU is touched: the index is the reference's own {Meta,FM} term and every read goes through the REFERENCE'S OWN
%%           sga:sga/6 and sw:sw/2 (its modules, called as they are -- nothing of them is restated here), folded exactly as
%%           worker_bwt.erl:84-97 folds them, so that both backends feed the same cast code below.  A worker without a GPU, or one
%%           cross-checking the GPU's casts, sets {backend, erlang}.
-module(worker_bwt_gpu).
-export([init_index/2, index_meta/1, running/4, workload/4]).

-define(REF_EXTENSION_LEN, 200).                                   %% worker_bwt.erl:17

%% replaces worker_bwt.erl:48-55: the index lives in HBM, Ref1 stays on the device as well
init_index(Chromosome, Device) ->
    {ok, FmIndices} = application:get_env(worker_bwt_app, fm_indices),
    Base = filename:join(FmIndices, Chromosome ++ "_p1"),          %% !Hardcoded part 1, as the reference
    case application:get_env(worker_bwt_app, backend, gpu) of
        gpu ->
            {ok, Index} = bwt_nif:index_open(Base ++ ".fm", Base ++ ".ref", Device),
            {gpu, Index};
        erlang ->
            {Meta, FM} = fm_index:get_index(Chromosome, 1),
            {ok, Ref} = file:read_file(Base ++ ".ref"),
            Ext = list_to_binary(lists:duplicate(?REF_EXTENSION_LEN, $N)),
            {erlang, Meta, FM, <<Ext/binary, Ref/binary, Ext/binary>>}
    end.

index_meta({gpu, Index}) -> bwt_nif:index_meta(Index);
index_meta({erlang, Meta, _FM, _Ref1}) ->
    {Pc,Pg,Pt,Last} = proplists:get_value(pointers, Meta),
    {Pc, Pg, Pt, Last, proplists:get_value(shift, Meta)}.

%% one QseqList -> [{Name, nothing | [{{Score,Cigar}, Pos, RefSeq}]}] in list order: what bwt_nif:align_batch/2 returns
align_batch({gpu, Index}, QseqList) -> bwt_nif:align_batch(Index, QseqList);
align_batch({erlang, Meta, FM, Ref1}, QseqList) ->
    {Pc,Pg,Pt,Last} = proplists:get_value(pointers, Meta),
    Shift = proplists:get_value(shift, Meta),
    [{Name, case sga:sga(FM, Pc, Pg, Pt, Last, Qsec) of
                [] -> nothing;
                Seeds ->
                    lists:foldl(fun({S, D}, Acc) ->
                        RefLen = length(Qsec) + D,
                        Start = S - RefLen + ?REF_EXTENSION_LEN,
                        <<_:Start/bytes, RefSeq:RefLen/bytes, _/binary>> = Ref1,
                        RefSeq1 = binary_to_list(RefSeq),
                        case sw:sw(Qsec, RefSeq1) of
                            no_match -> Acc;
                            Cigar -> [{Cigar, S - RefLen + Shift, RefSeq1} | Acc]
                        end
                    end, [], Seeds)
            end} || {Name, Qsec} <- QseqList].

%% A whole {workload, Workload} message (worker_bwt.erl:62-69) in ONE device call: bwt_nif:align_workload/2 answers the ten
%% QseqLists together (0.23 ms for 2000 reads where ten align_batch calls are 1.5 ms); the casts leave exactly as the reference
%% sends them -- list by list in message order (:73,112), inside a list in reverse read order (:74-80).
workload({gpu, Index}, Workload, MNode, MPid) ->
    lists:foreach(fun(Results) -> cast_results(Results, MNode, MPid) end, bwt_nif:align_workload(Index, Workload));
workload(Index, Workload, MNode, MPid) ->
    lists:foreach(fun(QseqList) -> running(Index, QseqList, MNode, MPid) end, Workload).

%% replaces worker_bwt.erl:73-106 for one QseqList
running(Index, QseqList, MNode, MPid) ->
    cast_results(align_batch(Index, QseqList), MNode, MPid).

cast_results(Results, MNode, MPid) ->
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
